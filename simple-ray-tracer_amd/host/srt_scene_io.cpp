/*
 * srt_scene_io.cpp -- scene input for the host mirror: Wavefront OBJ/MTL -> Mesh list, PNG -> Image, and the
 * prebuilt scene catalogue.  Triangle ids, vertex order and every coordinate bit feed the BVH build, so the
 * loader reproduces what the reference's loader produces:
 *   loadMesh / centerAndScaleToUnitMesh      reference framework/src/mesh.cpp:52-176
 *   Image::Image                             reference framework/src/image.cpp:13-35 (stb_image, 3 channels)
 *   loadScenePrebuilt / loadSceneFromFile    reference src/scene.cpp:5-105
 * The reference parses OBJ with tinyobjloader (vendored under framework/third_party/tinyobjloader, v2.0 line,
 * tiny_obj_loader.h); its published behaviour that matters here is restated: the hand-rolled decimal parser
 * (tiny_obj_loader.h:866-996: digits accumulated in double, fraction digits weighted by a power-of-ten table,
 * exponent applied as ldexp(m * 5^e, e), result narrowed to float), 1-based / negative relative indices
 * (:800-821), quads split along the shorter diagonal (:1428-1535), larger polygons ear-clipped in the
 * dominant plane (:1537-1819), `usemtl` changing only the per-face material id while `g` / `o` start a new
 * shape (:2690-2845), and the MTL reader (:1926-2315).
 */
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <numeric>
#include <sstream>
#include <unordered_map>

#include "srt_host.h"

namespace srt {

namespace {

inline vec3 operator+(vec3 a, vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline vec3 operator-(vec3 a, vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline vec3 operator*(vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3 operator/(vec3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 x, vec3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
inline float length(vec3 v) { return std::sqrt(dot(v, v)); }
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }

struct LoadError : std::exception {
    const char* what() const noexcept override { return "scene input failed"; }
};

/* ---------------------------------------------------------------- text helpers */

inline bool isSpace(char c) { return c == ' ' || c == '\t'; }
inline bool isDigit(char c) { return (unsigned)(c - '0') < 10u; }
inline bool isNewLine(char c) { return c == '\r' || c == '\n' || c == '\0'; }

/* Line reader that accepts \n, \r\n and a bare \r. */
bool getLine(std::istream& is, std::string& t)
{
    t.clear();
    std::streambuf* sb = is.rdbuf();
    if (!sb)
        return false;
    bool any = false;
    for (;;) {
        int c = sb->sbumpc();
        if (c == EOF) {
            if (!any)
                is.setstate(std::ios::eofbit);
            return any;
        }
        any = true;
        if (c == '\n')
            return true;
        if (c == '\r') {
            if (sb->sgetc() == '\n')
                sb->sbumpc();
            return true;
        }
        t += (char)c;
    }
}

/* Decimal parser with tinyobjloader's arithmetic (tiny_obj_loader.h:866-996). */
bool parseDouble(const char* s, const char* end, double* result)
{
    if (s >= end)
        return false;
    double mantissa = 0.0;
    int exponent = 0;
    char sign = '+', expSign = '+';
    const char* cur = s;
    int read = 0;
    bool leadingDot = false;
    if (*cur == '+' || *cur == '-') {
        sign = *cur++;
        if (cur != end && *cur == '.')
            leadingDot = true;
    } else if (isDigit(*cur)) {
    } else if (*cur == '.') {
        leadingDot = true;
    } else {
        return false;
    }
    bool more = cur != end;
    if (!leadingDot) {
        while (more && isDigit(*cur)) {
            mantissa *= 10;
            mantissa += (int)(*cur - '0');
            ++cur, ++read;
            more = cur != end;
        }
        if (read == 0)
            return false;
    }
    auto assemble = [&]() {
        *result = (sign == '+' ? 1 : -1) * (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
        return true;
    };
    if (!more)
        return assemble();
    if (*cur == '.') {
        ++cur;
        read = 1;
        more = cur != end;
        static const double lut[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
        while (more && isDigit(*cur)) {
            mantissa += (int)(*cur - '0') * (read < 8 ? lut[read] : std::pow(10.0, -read));
            ++read, ++cur;
            more = cur != end;
        }
    } else if (*cur == 'e' || *cur == 'E') {
    } else {
        return assemble();
    }
    if (!more)
        return assemble();
    if (*cur == 'e' || *cur == 'E') {
        ++cur;
        more = cur != end;
        if (more && (*cur == '+' || *cur == '-')) {
            expSign = *cur++;
        } else if (isDigit(*cur)) {
        } else {
            return false;
        }
        read = 0;
        more = cur != end;
        while (more && isDigit(*cur)) {
            if (exponent > 2147483647 / 10)
                return false;
            exponent *= 10;
            exponent += (int)(*cur - '0');
            ++cur, ++read;
            more = cur != end;
        }
        exponent *= (expSign == '+' ? 1 : -1);
        if (read == 0)
            return false;
    }
    return assemble();
}

float parseReal(const char*& tok, double def = 0.0)
{
    tok += std::strspn(tok, " \t");
    const char* end = tok + std::strcspn(tok, " \t\r");
    double v = def;
    parseDouble(tok, end, &v);
    tok = end;
    return (float)v;
}
bool parseRealOpt(const char*& tok, float* out)
{
    tok += std::strspn(tok, " \t");
    const char* end = tok + std::strcspn(tok, " \t\r");
    double v;
    bool ok = parseDouble(tok, end, &v);
    if (ok)
        *out = (float)v;
    tok = end;
    return ok;
}
std::string parseString(const char*& tok)
{
    tok += std::strspn(tok, " \t");
    size_t e = std::strcspn(tok, " \t\r");
    std::string s(tok, tok + e);
    tok += e;
    return s;
}

/* ---------------------------------------------------------------- MTL */

struct ObjMaterial {
    std::string name;
    float diffuse[3] = { 0, 0, 0 };
    float specular[3] = { 0, 0, 0 };
    float shininess = 1.0f;
    float dissolve = 1.0f;
    std::string diffuseTex;
};

/* Only the statements that reach a Material are interpreted; the texture statement keeps the
 * option-skipping of the original (a name is the first token that is not an option). */
std::string textureName(const char* tok)
{
    std::string name;
    while (!isNewLine(*tok)) {
        tok += std::strspn(tok, " \t");
        auto opt = [&](const char* o, int nargs) {
            size_t n = std::strlen(o);
            if (std::strncmp(tok, o, n) == 0 && isSpace(tok[n])) {
                tok += n + 1;
                for (int k = 0; k < nargs; ++k) {
                    tok += std::strspn(tok, " \t");
                    tok += std::strcspn(tok, " \t\r");
                }
                return true;
            }
            return false;
        };
        if (opt("-blendu", 1) || opt("-blendv", 1) || opt("-clamp", 1) || opt("-boost", 1) || opt("-bm", 1) || opt("-o", 3) || opt("-s", 3)
            || opt("-t", 3) || opt("-type", 1) || opt("-texres", 1) || opt("-imfchan", 1) || opt("-mm", 2) || opt("-colorspace", 1))
            continue;
        name = std::string(tok); /* rest of the line, spaces included */
        tok += name.size();
    }
    return name;
}

void loadMtl(std::map<std::string, int>& matMap, std::vector<ObjMaterial>& materials, std::istream& in)
{
    ObjMaterial cur;
    bool hasD = false, hasKd = false; /* hasKd is not reset per material in the original either */
    std::string line;
    while (in.peek() != -1) {
        getLine(in, line);
        if (!line.empty())
            line = line.substr(0, line.find_last_not_of(" \t") + 1);
        if (!line.empty() && line.back() == '\n') line.pop_back();
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty())
            continue;
        const char* tok = line.c_str();
        tok += std::strspn(tok, " \t");
        if (tok[0] == '\0' || tok[0] == '#')
            continue;
        if (std::strncmp(tok, "newmtl", 6) == 0 && isSpace(tok[6])) {
            if (!cur.name.empty()) {
                matMap.insert({ cur.name, (int)materials.size() });
                materials.push_back(cur);
            }
            cur = ObjMaterial();
            hasD = false;
            cur.name = std::string(tok + 7);
            continue;
        }
        if (tok[0] == 'K' && tok[1] == 'd' && isSpace(tok[2])) {
            tok += 2;
            for (float& c : cur.diffuse) c = parseReal(tok);
            hasKd = true;
            continue;
        }
        if (tok[0] == 'K' && tok[1] == 's' && isSpace(tok[2])) {
            tok += 2;
            for (float& c : cur.specular) c = parseReal(tok);
            continue;
        }
        if (tok[0] == 'N' && tok[1] == 's' && isSpace(tok[2])) {
            tok += 2;
            cur.shininess = parseReal(tok);
            continue;
        }
        if (tok[0] == 'd' && isSpace(tok[1])) {
            tok += 1;
            cur.dissolve = parseReal(tok);
            hasD = true;
            continue;
        }
        if (tok[0] == 'T' && tok[1] == 'r' && isSpace(tok[2])) {
            tok += 2;
            if (!hasD)
                cur.dissolve = 1.0f - parseReal(tok);
            continue;
        }
        if (std::strncmp(tok, "map_Kd", 6) == 0 && isSpace(tok[6])) {
            std::string n = textureName(tok + 7);
            if (!n.empty())
                cur.diffuseTex = n;
            if (!hasKd)
                cur.diffuse[0] = cur.diffuse[1] = cur.diffuse[2] = 0.6f;
            continue;
        }
    }
    matMap.insert({ cur.name, (int)materials.size() });
    materials.push_back(cur);
}

/* ---------------------------------------------------------------- OBJ */

struct Idx {
    int v = -1, vn = -1, vt = -1;
};

bool fixIndex(int idx, int n, int* ret)
{
    if (idx > 0) {
        *ret = idx - 1;
        return true;
    }
    if (idx == 0)
        return false;
    *ret = n + idx;
    return true;
}

bool parseTriple(const char*& tok, int vsize, int vnsize, int vtsize, Idx* out)
{
    Idx vi;
    if (!fixIndex(std::atoi(tok), vsize, &vi.v))
        return false;
    tok += std::strcspn(tok, "/ \t\r");
    if (tok[0] != '/') {
        *out = vi;
        return true;
    }
    ++tok;
    if (tok[0] == '/') { /* i//k */
        ++tok;
        if (!fixIndex(std::atoi(tok), vnsize, &vi.vn))
            return false;
        tok += std::strcspn(tok, "/ \t\r");
        *out = vi;
        return true;
    }
    if (!fixIndex(std::atoi(tok), vtsize, &vi.vt)) /* i/j or i/j/k */
        return false;
    tok += std::strcspn(tok, "/ \t\r");
    if (tok[0] != '/') {
        *out = vi;
        return true;
    }
    ++tok;
    if (!fixIndex(std::atoi(tok), vnsize, &vi.vn))
        return false;
    tok += std::strcspn(tok, "/ \t\r");
    *out = vi;
    return true;
}

struct Shape {
    std::vector<Idx> indices;     /* three per triangle */
    std::vector<int> materialIds; /* one per triangle */
};

/* Crossing-number point-in-polygon test used by the ear clipper. */
int pnpoly(int n, const float* vx, const float* vy, float tx, float ty)
{
    int c = 0;
    for (int i = 0, j = n - 1; i < n; j = i++)
        if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i]))
            c = !c;
    return c;
}

/* Triangulates the pending faces into the shape (tiny_obj_loader.h:1401-1840). */
void flushFaces(Shape& shape, std::vector<std::vector<Idx>>& faces, int materialId, const std::vector<float>& v)
{
    auto emit = [&](const Idx& a, const Idx& b, const Idx& c) {
        shape.indices.push_back(a), shape.indices.push_back(b), shape.indices.push_back(c);
        shape.materialIds.push_back(materialId);
    };
    for (const std::vector<Idx>& face : faces) {
        size_t np = face.size();
        if (np < 3)
            continue;
        if (np == 4) {
            const Idx &i0 = face[0], &i1 = face[1], &i2 = face[2], &i3 = face[3];
            size_t a = (size_t)i0.v, b = (size_t)i1.v, c = (size_t)i2.v, d = (size_t)i3.v;
            if (3 * a + 2 >= v.size() || 3 * b + 2 >= v.size() || 3 * c + 2 >= v.size() || 3 * d + 2 >= v.size())
                continue;
            float e02x = v[c * 3] - v[a * 3], e02y = v[c * 3 + 1] - v[a * 3 + 1], e02z = v[c * 3 + 2] - v[a * 3 + 2];
            float e13x = v[d * 3] - v[b * 3], e13y = v[d * 3 + 1] - v[b * 3 + 1], e13z = v[d * 3 + 2] - v[b * 3 + 2];
            float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
            float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
            if (sqr02 < sqr13) {
                emit(i0, i1, i2);
                emit(i0, i2, i3);
            } else {
                emit(i0, i1, i3);
                emit(i1, i2, i3);
            }
            continue;
        }
        /* projection axes: drop the dominant component of the first non-degenerate corner */
        size_t axes[2] = { 1, 2 };
        for (size_t k = 0; k < np; ++k) {
            size_t a = (size_t)face[k % np].v, b = (size_t)face[(k + 1) % np].v, c = (size_t)face[(k + 2) % np].v;
            if (3 * a + 2 >= v.size() || 3 * b + 2 >= v.size() || 3 * c + 2 >= v.size())
                continue;
            float e0x = v[b * 3] - v[a * 3], e0y = v[b * 3 + 1] - v[a * 3 + 1], e0z = v[b * 3 + 2] - v[a * 3 + 2];
            float e1x = v[c * 3] - v[b * 3], e1y = v[c * 3 + 1] - v[b * 3 + 1], e1z = v[c * 3 + 2] - v[b * 3 + 2];
            float cx = std::fabs(e0y * e1z - e0z * e1y), cy = std::fabs(e0z * e1x - e0x * e1z), cz = std::fabs(e0x * e1y - e0y * e1x);
            const float eps = std::numeric_limits<float>::epsilon();
            if (cx > eps || cy > eps || cz > eps) {
                if (!(cx > cy && cx > cz)) {
                    axes[0] = 0;
                    if (cz > cx && cz > cy)
                        axes[1] = 1;
                }
                break;
            }
        }
        std::vector<Idx> rem = face;
        size_t guess = 0, remainingIter = face.size(), prevRemaining = rem.size();
        while (rem.size() > 3 && remainingIter > 0) {
            np = rem.size();
            if (guess >= np)
                guess -= np;
            if (prevRemaining != np) {
                prevRemaining = np;
                remainingIter = np;
            } else {
                remainingIter--;
            }
            Idx ind[3];
            float vx[3], vy[3];
            for (size_t k = 0; k < 3; ++k) {
                ind[k] = rem[(guess + k) % np];
                size_t vi = (size_t)ind[k].v;
                if (vi * 3 + axes[0] >= v.size() || vi * 3 + axes[1] >= v.size())
                    vx[k] = vy[k] = 0.0f;
                else
                    vx[k] = v[vi * 3 + axes[0]], vy[k] = v[vi * 3 + axes[1]];
            }
            float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
            float crs = e0x * e1y - e0y * e1x;
            float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
            if (crs * area < 0.0f) {
                guess += 1;
                continue;
            }
            bool overlap = false;
            for (size_t other = 3; other < np; ++other) {
                size_t idx = (guess + other) % np;
                if (idx >= rem.size())
                    continue;
                size_t ovi = (size_t)rem[idx].v;
                if (ovi * 3 + axes[0] >= v.size() || ovi * 3 + axes[1] >= v.size())
                    continue;
                if (pnpoly(3, vx, vy, v[ovi * 3 + axes[0]], v[ovi * 3 + axes[1]])) {
                    overlap = true;
                    break;
                }
            }
            if (overlap) {
                guess += 1;
                continue;
            }
            emit(ind[0], ind[1], ind[2]);
            size_t removed = (guess + 1) % np;
            while (removed + 1 < np) {
                rem[removed] = rem[removed + 1];
                removed += 1;
            }
            rem.pop_back();
        }
        if (rem.size() == 3)
            emit(rem[0], rem[1], rem[2]);
    }
    faces.clear();
}

struct ObjFile {
    std::vector<float> v, vn, vt;
    std::vector<Shape> shapes;
    std::vector<ObjMaterial> materials;
};

bool loadObj(ObjFile& out, const std::string& filename, const std::string& mtlBaseDir)
{
    std::ifstream ifs(filename);
    if (!ifs)
        return false;
    std::string baseDir = mtlBaseDir;
    if (!baseDir.empty() && baseDir.back() != '/')
        baseDir += '/';
    std::map<std::string, int> matMap;
    int material = -1;
    Shape shape;
    std::vector<std::vector<Idx>> faces;
    std::string line;
    while (ifs.peek() != -1) {
        getLine(ifs, line);
        if (!line.empty() && line.back() == '\n') line.pop_back();
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty())
            continue;
        const char* tok = line.c_str();
        tok += std::strspn(tok, " \t");
        if (tok[0] == '\0' || tok[0] == '#')
            continue;
        if (tok[0] == 'v' && isSpace(tok[1])) {
            tok += 2;
            float x = parseReal(tok), y = parseReal(tok), z = parseReal(tok);
            float r, g, b; /* optional vertex colour: parsed and dropped */
            (void)(parseRealOpt(tok, &r) && parseRealOpt(tok, &g) && parseRealOpt(tok, &b));
            out.v.push_back(x), out.v.push_back(y), out.v.push_back(z);
            continue;
        }
        if (tok[0] == 'v' && tok[1] == 'n' && isSpace(tok[2])) {
            tok += 3;
            float x = parseReal(tok), y = parseReal(tok), z = parseReal(tok);
            out.vn.push_back(x), out.vn.push_back(y), out.vn.push_back(z);
            continue;
        }
        if (tok[0] == 'v' && tok[1] == 't' && isSpace(tok[2])) {
            tok += 3;
            float x = parseReal(tok), y = parseReal(tok);
            out.vt.push_back(x), out.vt.push_back(y);
            continue;
        }
        if (tok[0] == 'f' && isSpace(tok[1])) {
            tok += 2;
            tok += std::strspn(tok, " \t");
            std::vector<Idx> face;
            while (!isNewLine(tok[0])) {
                Idx vi;
                if (!parseTriple(tok, (int)(out.v.size() / 3), (int)(out.vn.size() / 3), (int)(out.vt.size() / 2), &vi))
                    return false;
                face.push_back(vi);
                tok += std::strspn(tok, " \t\r");
            }
            faces.push_back(std::move(face));
            continue;
        }
        if (std::strncmp(tok, "usemtl", 6) == 0) {
            tok += 6;
            std::string name = parseString(tok);
            int newId = -1;
            auto it = matMap.find(name);
            if (it != matMap.end())
                newId = it->second;
            if (newId != material) {
                flushFaces(shape, faces, material, out.v);
                material = newId;
            }
            continue;
        }
        if (std::strncmp(tok, "mtllib", 6) == 0 && isSpace(tok[6])) {
            tok += 7;
            /* file names separated by blanks, backslash escapes a blank; the first that opens is used */
            std::vector<std::string> names;
            std::string curName;
            bool esc = false;
            for (const char* c = tok; *c; ++c) {
                if (esc) {
                    curName += *c;
                    esc = false;
                } else if (*c == '\\') {
                    esc = true;
                } else if (*c == ' ') {
                    if (!curName.empty())
                        names.push_back(curName);
                    curName.clear();
                } else {
                    curName += *c;
                }
            }
            if (!curName.empty())
                names.push_back(curName);
            for (const std::string& n : names) {
                std::ifstream ms(baseDir.empty() ? n : baseDir + n);
                if (ms) {
                    loadMtl(matMap, out.materials, ms);
                    break;
                }
            }
            continue;
        }
        if ((tok[0] == 'g' || tok[0] == 'o') && isSpace(tok[1])) {
            flushFaces(shape, faces, material, out.v);
            if (!shape.indices.empty())
                out.shapes.push_back(shape);
            shape = Shape();
            continue;
        }
    }
    const bool hadFaces = !faces.empty();
    flushFaces(shape, faces, material, out.v);
    if (hadFaces || !shape.indices.empty())
        out.shapes.push_back(shape);
    return true;
}

/* Vertex identity for de-duplication: float equality per component (so -0 == 0), first occurrence wins. */
struct VertexKey {
    Vertex v;
    bool operator==(const VertexKey& o) const
    {
        return v.position.x == o.v.position.x && v.position.y == o.v.position.y && v.position.z == o.v.position.z && v.normal.x == o.v.normal.x
            && v.normal.y == o.v.normal.y && v.normal.z == o.v.normal.z && v.texCoord.x == o.v.texCoord.x && v.texCoord.y == o.v.texCoord.y;
    }
};
struct VertexKeyHash {
    size_t operator()(const VertexKey& k) const
    {
        const float f[8] = { k.v.position.x, k.v.position.y, k.v.position.z, k.v.normal.x, k.v.normal.y, k.v.normal.z, k.v.texCoord.x, k.v.texCoord.y };
        size_t seed = 0;
        for (float x : f) {
            uint32_t b;
            float y = (x == 0.0f) ? 0.0f : x;
            std::memcpy(&b, &y, 4);
            seed ^= (size_t)b + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        }
        return seed;
    }
};

/* centerAndScaleToUnitMesh, reference framework/src/mesh.cpp:151-176 */
void centerAndScaleToUnit(std::vector<Mesh>& meshes)
{
    std::vector<vec3> positions;
    for (const Mesh& m : meshes)
        for (const Vertex& v : m.vertices)
            positions.push_back(v.position);
    vec3 sum(0.0f);
    for (const vec3& p : positions)
        sum = sum + p;
    const vec3 center = sum / static_cast<float>(positions.size());
    float maxD = 0.0f;
    for (const vec3& p : positions)
        maxD = std::max(length(p - center), maxD);
    for (Mesh& m : meshes)
        for (Vertex& v : m.vertices)
            v.position = (v.position - center) / maxD;
}

/* ---------------------------------------------------------------- PNG */

uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc)
        return a;
    if (pb <= pc)
        return b;
    return c;
}

/* PNG -> RGB8, matching what stb_image returns for req_comp = 3: palette expanded, grey replicated (scaled
 * to 0..255 for depths below 8), alpha dropped, 16-bit samples reduced to their high byte. */
bool decodePng(const std::vector<uint8_t>& file, int& w, int& h, std::vector<uint8_t>& rgb)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0)
        return false;
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool gotHeader = false;
    while (pos + 12 <= file.size()) {
        uint32_t len = be32(&file[pos]);
        const char* type = (const char*)&file[pos + 4];
        const uint8_t* data = &file[pos + 8];
        if (pos + 12 + len > file.size())
            return false;
        if (!std::strncmp(type, "IHDR", 4)) {
            if (len < 13)
                return false;
            w = (int)be32(data), h = (int)be32(data + 4);
            depth = data[8], ctype = data[9], interlace = data[12];
            gotHeader = true;
        } else if (!std::strncmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!std::strncmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::strncmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    if (!gotHeader || w <= 0 || h <= 0 || interlace > 1)
        return false;
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels || (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16))
        return false;
    const size_t bitsPerPixel = (size_t)channels * depth;
    const size_t bpp = std::max<size_t>(1, bitsPerPixel / 8);
    /* The image is stored as one pass, or (Adam7 interlacing) as seven reduced images: pass p holds the pixels
     * (x0 + i * dx, y0 + j * dy), each pass filtered like an image of its own. */
    struct Pass {
        int x0, y0, dx, dy;
    };
    static const Pass adam7[7] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
    static const Pass whole = { 0, 0, 1, 1 };
    const int nPasses = interlace ? 7 : 1;
    size_t rawSize = 0;
    for (int p = 0; p < nPasses; ++p) {
        const Pass& ps = interlace ? adam7[p] : whole;
        const size_t pw = (size_t)(w - ps.x0 + ps.dx - 1) / ps.dx, ph = (size_t)(h - ps.y0 + ps.dy - 1) / ps.dy;
        if (w > ps.x0 && h > ps.y0 && pw && ph)
            rawSize += ((pw * bitsPerPixel + 7) / 8 + 1) * ph;
    }
    /* deflate expands by at most 1032 : 1, so a header that promises more than the data can hold is refused
     * before anything of that size is allocated (the length check after uncompress would refuse it anyway) */
    if (w > (1 << 24) || h > (1 << 24) || rawSize > idat.size() * 1032 + 1024)
        return false;
    std::vector<uint8_t> raw(rawSize);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size())
        return false;
    rgb.assign((size_t)w * h * 3, 0);
    static const int depthScale[9] = { 0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01 };
    size_t rawPos = 0;
    for (int p = 0; p < nPasses; ++p) {
        const Pass& ps = interlace ? adam7[p] : whole;
        if (w <= ps.x0 || h <= ps.y0)
            continue;
        const int pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        const size_t stride = ((size_t)pw * bitsPerPixel + 7) / 8;
        std::vector<uint8_t> img(stride * (size_t)ph);
        for (int y = 0; y < ph; ++y) {
            const uint8_t* in = &raw[rawPos + (stride + 1) * (size_t)y];
            const int filter = in[0];
            ++in;
            uint8_t* cur = &img[stride * (size_t)y];
            const uint8_t* up = y ? &img[stride * (size_t)(y - 1)] : nullptr;
            for (size_t i = 0; i < stride; ++i) {
                int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
                int x = in[i];
                switch (filter) {
                case 0: break;
                case 1: x += a; break;
                case 2: x += b; break;
                case 3: x += (a + b) >> 1; break;
                case 4: x += paeth(a, b, c); break;
                default: return false;
                }
                cur[i] = (uint8_t)x;
            }
        }
        rawPos += (stride + 1) * (size_t)ph;
        for (int y = 0; y < ph; ++y) {
            const uint8_t* row = &img[stride * (size_t)y];
            for (int x = 0; x < pw; ++x) {
                auto sample = [&](int ch) -> int {
                    if (depth == 16)
                        return row[((size_t)x * channels + ch) * 2];
                    if (depth == 8)
                        return row[(size_t)x * channels + ch];
                    size_t bit = ((size_t)x * channels + ch) * depth;
                    return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                };
                uint8_t* o = &rgb[((size_t)(ps.y0 + y * ps.dy) * w + (size_t)(ps.x0 + x * ps.dx)) * 3];
                if (ctype == 3) {
                    size_t idx = (size_t)sample(0);
                    if (idx * 3 + 2 >= plte.size())
                        return false;
                    o[0] = plte[idx * 3], o[1] = plte[idx * 3 + 1], o[2] = plte[idx * 3 + 2];
                } else if (ctype == 0 || ctype == 4) {
                    int g = sample(0);
                    if (depth < 8)
                        g *= depthScale[depth];
                    o[0] = o[1] = o[2] = (uint8_t)g;
                } else {
                    o[0] = (uint8_t)sample(0), o[1] = (uint8_t)sample(1), o[2] = (uint8_t)sample(2);
                }
            }
        }
    }
    return true;
}

} // namespace

/* srt_jpeg.cpp */
bool decodeJpeg(const std::vector<uint8_t>& file, int& w, int& h, std::vector<uint8_t>& rgb);

/* ------------------------------------------------------------------------------------------ public */

Image::Image(const std::filesystem::path& filePath)
{
    if (!std::filesystem::exists(filePath)) {
        std::cerr << "Texture file " << filePath << " does not exists!" << std::endl;
        throw LoadError();
    }
    std::ifstream in(filePath, std::ios::binary);
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    std::vector<uint8_t> rgb;
    if (!decodePng(bytes, width, height, rgb) && !decodeJpeg(bytes, width, height, rgb)) {
        std::cerr << "Failed to read texture " << filePath << " (PNG and Huffman-coded 8-bit JPEG files are decoded)" << std::endl;
        throw LoadError();
    }
    pixels.reserve((size_t)width * height);
    for (size_t i = 0; i < rgb.size(); i += 3)
        pixels.emplace_back(rgb[i] / 255.0f, rgb[i + 1] / 255.0f, rgb[i + 2] / 255.0f);
}

std::shared_ptr<Image> loadEnvironment(const std::filesystem::path& dataDir) { return std::make_shared<Image>(dataDir / "cube_map.jpg"); }

/* Triangle ranges [first, last) of a shape's sub-meshes: one per run of equal material ids, where a boundary is only
 * looked for at triangles 1 .. n-2 - a last triangle with a material of its own stays in the run before it, as in the
 * reference's splitting loop (reference framework/src/mesh.cpp:76-85, 141). */
static std::vector<std::pair<size_t, size_t>> materialRuns(const std::vector<int>& ids, size_t n)
{
    std::vector<std::pair<size_t, size_t>> runs;
    if (n == 0)
        return runs;
    size_t first = 0;
    for (size_t t = 1; t + 1 < n; ++t)
        if (ids[t] != ids[t - 1]) {
            runs.emplace_back(first, t);
            first = t;
        }
    runs.emplace_back(first, n);
    return runs;
}

/* One sub-mesh: corners become vertices in first-seen order, a corner equal to an earlier one in position, normal and
 * texture coordinate shares its vertex; corners without a normal in the file take the triangle's geometric normal
 * (reference framework/src/mesh.cpp:87-137). */
static Mesh subMesh(const ObjFile& obj, const Shape& shape, size_t first, size_t last, const std::filesystem::path& baseDir)
{
    auto position = [&](const Idx& c) { return vec3(obj.v[3 * (size_t)c.v], obj.v[3 * (size_t)c.v + 1], obj.v[3 * (size_t)c.v + 2]); };
    const bool fileHasNormals = !obj.vn.empty(), fileHasUvs = !obj.vt.empty();
    Mesh mesh;
    std::unordered_map<VertexKey, uint32_t, VertexKeyHash> seen;
    for (size_t t = first; t < last; ++t) {
        const Idx* corner = &shape.indices[3 * t];
        const vec3 p0 = position(corner[0]), p1 = position(corner[1]), p2 = position(corner[2]);
        const vec3 faceNormal = normalize(cross(p1 - p0, p2 - p0));
        uint32_t index[3];
        for (int k = 0; k < 3; ++k) {
            const Idx& c = corner[k];
            Vertex v;
            v.position = position(c);
            v.normal = (c.vn != -1 && fileHasNormals) ? vec3(obj.vn[3 * (size_t)c.vn], obj.vn[3 * (size_t)c.vn + 1], obj.vn[3 * (size_t)c.vn + 2]) : faceNormal;
            if (c.vt != -1 && fileHasUvs)
                v.texCoord = { obj.vt[2 * (size_t)c.vt], obj.vt[2 * (size_t)c.vt + 1] };
            const auto [slot, fresh] = seen.try_emplace(VertexKey { v }, (uint32_t)mesh.vertices.size());
            if (fresh)
                mesh.vertices.push_back(v);
            index[k] = slot->second;
        }
        mesh.triangles.push_back({ index[0], index[1], index[2] });
    }
    const int id = shape.materialIds[first];
    if (id < 0) { /* no material in the file: white, diffuse only */
        mesh.material.kd = vec3(1.0f);
        mesh.material.ks = vec3(0.0f);
        mesh.material.shininess = 1.0f;
    } else {
        const ObjMaterial& m = obj.materials[(size_t)id];
        mesh.material.kd = vec3(m.diffuse[0], m.diffuse[1], m.diffuse[2]);
        mesh.material.ks = vec3(m.specular[0], m.specular[1], m.specular[2]);
        mesh.material.shininess = m.shininess;
        mesh.material.transparency = m.dissolve;
        if (!m.diffuseTex.empty())
            mesh.material.kdTexture = std::make_shared<Image>(baseDir / m.diffuseTex);
    }
    return mesh;
}

std::vector<Mesh> loadMesh(const std::filesystem::path& file, bool centerAndNormalize)
{
    if (!std::filesystem::exists(file)) {
        std::cerr << "File " << file << " does not exist." << std::endl;
        throw LoadError();
    }
    const auto baseDir = file.parent_path();
    ObjFile obj;
    if (!loadObj(obj, file.string(), baseDir.string())) {
        std::cerr << "Failed to load mesh " << file << std::endl;
        throw LoadError();
    }
    /* Indices that point outside the file's arrays are undefined behaviour in the reference's loader; here the file
     * is refused. */
    const long nV = (long)(obj.v.size() / 3), nVn = (long)(obj.vn.size() / 3), nVt = (long)(obj.vt.size() / 2), nMat = (long)obj.materials.size();
    for (const Shape& shape : obj.shapes) {
        for (const Idx& ix : shape.indices)
            if (ix.v < 0 || ix.v >= nV || ix.vn < -1 || (nVn && ix.vn >= nVn) || ix.vt < -1 || (nVt && ix.vt >= nVt)) { /* an empty normal / uv array is never indexed */
                std::cerr << "Mesh " << file << " refers to a vertex, normal or texture coordinate it does not define" << std::endl;
                throw LoadError();
            }
        for (int id : shape.materialIds)
            if (id < -1 || id >= nMat) {
                std::cerr << "Mesh " << file << " refers to a material it does not define" << std::endl;
                throw LoadError();
            }
    }
    std::vector<Mesh> out;
    for (const Shape& shape : obj.shapes)
        for (const auto& [first, last] : materialRuns(shape.materialIds, shape.indices.size() / 3))
            out.push_back(subMesh(obj, shape, first, last, baseDir));
    if (centerAndNormalize)
        centerAndScaleToUnit(out);
    return out;
}

Scene loadScenePrebuilt(SceneType type, const std::filesystem::path& dataDir)
{
    Scene scene;
    scene.type = type;
    auto add = [&](const char* file, bool normalize) {
        auto sub = loadMesh(dataDir / file, normalize);
        for (Mesh& m : sub)
            scene.meshes.push_back(std::move(m));
    };
    switch (type) {
    case SingleTriangle:
        add("triangle.obj", false);
        scene.meshes[0].material.kd = vec3(1.0f);
        scene.lights.emplace_back(PointLight { vec3(-1, 1, -1), vec3(1) });
        break;
    case Cube:
        add("cube.obj", false);
        scene.lights.emplace_back(SegmentLight { vec3(1.5f, 0.5f, -0.6f), vec3(-1, 0.5f, -0.5f), vec3(0.9f, 0.2f, 0.1f), vec3(0.2f, 1, 0.3f) });
        break;
    case CubeTextured:
        add("cube-textured.obj", false);
        scene.lights.emplace_back(PointLight { vec3(-1.0f, 1.5f, -1.0f), vec3(1) });
        break;
    case CornellBox:
        add("CornellBox-Mirror-Rotated.obj", true);
        scene.lights.emplace_back(PointLight { vec3(0, 0.58f, 0), vec3(1) });
        break;
    case CornellBoxParallelogramLight:
        add("CornellBox-Mirror-Rotated.obj", true);
        scene.lights.emplace_back(ParallelogramLight { vec3(-0.2f, 0.5f, 0), vec3(0.4f, 0, 0), vec3(0.0f, 0.0f, 0.4f), vec3(1, 0, 0), vec3(0, 1, 0),
            vec3(0, 0, 1), vec3(0, 1, 1) });
        break;
    case Monkey:
        add("monkey.obj", true);
        scene.lights.emplace_back(PointLight { vec3(-1, 1, -1), vec3(1) });
        scene.lights.emplace_back(PointLight { vec3(1, -1, -1), vec3(1) });
        break;
    case Teapot:
        add("teapot.obj", true);
        scene.lights.emplace_back(PointLight { vec3(-1, 1, -1), vec3(1) });
        break;
    case Dragon:
        add("dragon.obj", true);
        scene.lights.emplace_back(PointLight { vec3(-1, 1, -1), vec3(1) });
        break;
    case Spheres:
        /* the reference's sphere intersection always misses (src/intersect.cpp:81-125): spheres are carried, never hit */
        scene.spheres.push_back(Sphere { vec3(3.0f, -2.0f, 10.2f), 1.0f, Material { vec3(0.8f, 0.2f, 0.2f) } });
        scene.spheres.push_back(Sphere { vec3(-2.0f, 2.0f, 4.0f), 2.0f, Material { vec3(0.6f, 0.8f, 0.2f) } });
        scene.spheres.push_back(Sphere { vec3(0.0f, 0.0f, 6.0f), 0.75f, Material { vec3(0.2f, 0.2f, 0.8f) } });
        scene.lights.emplace_back(PointLight { vec3(3, 0, 3), vec3(15) });
        break;
    case Custom:
        add("custom.obj", false);
        scene.lights.emplace_back(PointLight { vec3(-1, 1, -1), vec3(1) });
        break;
    }
    return scene;
}

Scene loadSceneFromFile(const std::filesystem::path& path, const std::vector<Light>& lights)
{
    Scene scene;
    scene.lights = lights;
    auto sub = loadMesh(path);
    for (Mesh& m : sub)
        scene.meshes.push_back(std::move(m));
    return scene;
}

} // namespace srt
