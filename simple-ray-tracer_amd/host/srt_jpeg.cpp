/*
 * srt_jpeg.cpp -- JPEG decoder of the host mirror's Image loader.
 *
 * The reference loads images through stb_image (framework/src/image.cpp:13-35, three channels), and its
 * environment map is a progressive JPEG (data/cube_map.jpg), so the pixels its renderer looks up are whatever
 * stb_image's decoder makes of the file.  JPEG leaves the arithmetic of the inverse DCT, the chroma upsampling and the
 * colour conversion to the decoder, so "the same pixels" means restating stb_image's choices:
 *   - coefficients kept as 16-bit integers (dequantisation wraps at 16 bits),
 *   - the integer inverse DCT derived from the IJG "islow" one: constants scaled by 4096, columns first with two
 *     extra bits kept (+512 >> 10), then rows (+65536 + (128 << 17) >> 17), clamped to 0..255,
 *   - "fancy" upsampling for factors of two: 3/4 near + 1/4 far with rounding, horizontally, vertically, or both
 *     (3 * (3 near + far) + (3 near' + far') + 8 >> 4); nearest neighbour for any other factor,
 *   - YCbCr -> RGB in 20-bit fixed point with the constants rounded to 12 bits first, and the Cb term of green cut
 *     to its upper 16 bits,
 *   - RGB passthrough for component ids 'R','G','B' or an Adobe segment with transform 0 and no JFIF segment; CMYK /
 *     YCCK through an 8-bit rounded multiply.
 * Baseline and progressive Huffman files with 8-bit samples are read (as by stb_image); arithmetic coding, lossless
 * and 12-bit files are refused.  Checked pixel for pixel against the reference loader's output on generated files and
 * on cube_map.jpg (tests/golden/image_cases, made by oracle/make_golden.py).
 */
#include <cstdint>
#include <cstring>
#include <vector>

namespace srt {

namespace {

const uint8_t kZigzag[64 + 16] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
    28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    /* a corrupt run may step past the block: those writes land on the last coefficient */
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63 };

struct Huffman {
    bool defined = false;
    uint8_t values[256] = {};
    int32_t maxCode[18] = {}; /* largest code of each length, -1 when the length has none */
    int32_t firstCode[17] = {};
    int32_t firstIndex[17] = {};
    int16_t fast[512]; /* 9-bit prefix -> (length << 8 | symbol), -1 when the code is longer */

    bool build(const int counts[16], const uint8_t* vals, int n)
    {
        std::memcpy(values, vals, (size_t)n);
        int code = 0, index = 0;
        for (int i = 0; i < 512; ++i)
            fast[i] = -1;
        for (int len = 1; len <= 16; ++len) {
            firstCode[len] = code;
            firstIndex[len] = index;
            for (int k = 0; k < counts[len - 1]; ++k, ++code, ++index) {
                if (code >= (1 << len))
                    return false; /* more codes than the length can hold */
                if (len <= 9)
                    for (int f = 0; f < (1 << (9 - len)); ++f)
                        fast[(code << (9 - len)) | f] = (int16_t)((len << 8) | values[index]);
            }
            maxCode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        defined = true;
        return true;
    }
};

/* Entropy-coded segment reader: bits most significant first, 0xFF 0x00 is a data byte 0xFF, any other 0xFF xx is a
 * marker, which ends the data (zeros are fed from there on) and is remembered. */
struct BitReader {
    const uint8_t* data;
    size_t size, pos;
    uint32_t buffer = 0;
    int bits = 0;
    int marker = -1;

    uint8_t byteAt(size_t i) const { return i < size ? data[i] : 0; }
    void refill()
    {
        while (bits <= 24) {
            uint32_t b = 0;
            if (marker < 0) {
                b = byteAt(pos++);
                if (b == 0xFF) {
                    uint8_t c = byteAt(pos++);
                    while (c == 0xFF)
                        c = byteAt(pos++);
                    if (c != 0) {
                        marker = c;
                        b = 0; /* zeros from here on, so that the bit count never falls below what a caller drops */
                    }
                }
            }
            buffer |= b << (24 - bits);
            bits += 8;
        }
    }
    uint32_t peek16()
    {
        if (bits < 16)
            refill();
        return buffer >> 16;
    }
    void drop(int n)
    {
        buffer <<= n;
        bits -= n;
    }
    int get(int n)
    {
        if (n == 0)
            return 0;
        if (bits < n)
            refill();
        const int v = (int)(buffer >> (32 - n));
        drop(n);
        return v;
    }
    void reset()
    {
        buffer = 0, bits = 0, marker = -1;
    }
};

int decodeSymbol(BitReader& br, const Huffman& h)
{
    const uint32_t look = br.peek16();
    const int f = h.fast[look >> 7];
    if (f >= 0) {
        if ((f >> 8) > br.bits)
            return -1;
        br.drop(f >> 8);
        return f & 255;
    }
    for (int len = 10; len <= 16; ++len) {
        const int code = (int)(look >> (16 - len));
        if (h.maxCode[len] >= 0 && code <= h.maxCode[len] && code >= h.firstCode[len]) {
            if (len > br.bits)
                return -1;
            br.drop(len);
            return h.values[h.firstIndex[len] + code - h.firstCode[len]];
        }
    }
    return -1;
}

/* the JPEG EXTEND of an n-bit magnitude */
int receiveExtend(BitReader& br, int n)
{
    if (n == 0)
        return 0;
    const int v = br.get(n);
    return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0, dcPred = 0;
    std::vector<uint8_t> pixels; /* w2 x h2 */
    std::vector<int16_t> coeff;  /* progressive: 64 per block, w2 / 8 blocks per row */
};

inline uint8_t clampByte(int x) { return x < 0 ? 0 : (x > 255 ? 255 : (uint8_t)x); }

/* One pass of the 8-point inverse DCT on (s0..s7); leaves the even part in x0..x3 and the odd part in t0..t3. */
#define SRT_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                                                \
    int32_t t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                                     \
    p2 = s2, p3 = s6;                                                                                               \
    p1 = (p2 + p3) * 2217;                                                                                          \
    t2 = p1 + p3 * -7567;                                                                                           \
    t3 = p1 + p2 * 3135;                                                                                            \
    p2 = s0, p3 = s4;                                                                                               \
    t0 = (p2 + p3) * 4096;                                                                                          \
    t1 = (p2 - p3) * 4096;                                                                                          \
    x0 = t0 + t3, x3 = t0 - t3, x1 = t1 + t2, x2 = t1 - t2;                                                         \
    t0 = s7, t1 = s5, t2 = s3, t3 = s1;                                                                             \
    p3 = t0 + t2, p4 = t1 + t3, p1 = t0 + t3, p2 = t1 + t2;                                                         \
    p5 = (p3 + p4) * 4816;                                                                                          \
    t0 = t0 * 1223, t1 = t1 * 8410, t2 = t2 * 12586, t3 = t3 * 6149;                                                \
    p1 = p5 + p1 * -3685;                                                                                           \
    p2 = p5 + p2 * -10497;                                                                                          \
    p3 = p3 * -8034;                                                                                                \
    p4 = p4 * -1597;                                                                                                \
    t3 += p1 + p4, t2 += p2 + p3, t1 += p2 + p4, t0 += p1 + p3;

/* The constants above are (int)(c * 4096 + 0.5) of 0.5411961, -1.847759065, 0.765366865, 1.175875602, 0.298631336,
 * 2.053119869, 3.072711026, 1.501321110, -0.899976223, -2.562915447, -1.961570560, -0.390180644 (the cast truncates
 * towards zero, hence -7567, -3685, -10497, -8034, -1597). */
void inverseDct(uint8_t* out, int stride, const int16_t d[64])
{
    int32_t val[64];
    for (int i = 0; i < 8; ++i) {
        const int16_t* c = d + i;
        int32_t* v = val + i;
        if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
            const int32_t dc = c[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
        } else {
            SRT_IDCT_1D(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56])
            x0 += 512, x1 += 512, x2 += 512, x3 += 512;
            v[0] = (x0 + t3) >> 10, v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10, v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10, v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10, v[32] = (x3 - t0) >> 10;
        }
    }
    for (int i = 0; i < 8; ++i) {
        const int32_t* v = val + 8 * i;
        uint8_t* o = out + (size_t)stride * i;
        SRT_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        const int32_t bias = 65536 + (128 << 17);
        x0 += bias, x1 += bias, x2 += bias, x3 += bias;
        o[0] = clampByte((x0 + t3) >> 17), o[7] = clampByte((x0 - t3) >> 17);
        o[1] = clampByte((x1 + t2) >> 17), o[6] = clampByte((x1 - t2) >> 17);
        o[2] = clampByte((x2 + t1) >> 17), o[5] = clampByte((x2 - t1) >> 17);
        o[3] = clampByte((x3 + t0) >> 17), o[4] = clampByte((x3 - t0) >> 17);
    }
}

struct Decoder {
    const uint8_t* data;
    size_t size, pos = 0;
    int width = 0, height = 0, nComp = 0;
    bool progressive = false, jfif = false;
    int adobeTransform = -1, rgbIds = 0;
    int hMax = 1, vMax = 1, mcuX = 0, mcuY = 0;
    int restartInterval = 0;
    uint16_t dequant[4][64] = {};
    Huffman dcTables[4], acTables[4];
    Component comp[4];
    /* current scan */
    int scanN = 0, order[4] = {}, specStart = 0, specEnd = 63, succHigh = 0, succLow = 0, eobRun = 0;
    int pendingMarker = -1;

    int get8() { return pos < size ? data[pos++] : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }
    bool atEnd() const { return pos >= size; }

    /* next marker code, or -1 when the next byte does not start one */
    int nextMarker()
    {
        if (pendingMarker >= 0) {
            const int m = pendingMarker;
            pendingMarker = -1;
            return m;
        }
        int x = get8();
        if (x != 0xFF)
            return -1;
        while (x == 0xFF)
            x = get8();
        return x;
    }

    bool tablesAndSegments(int m)
    {
        if (m < 0)
            return false;
        if (m == 0xDD) {
            if (get16() != 4)
                return false;
            restartInterval = get16();
            return true;
        }
        if (m == 0xDB) {
            int L = get16() - 2;
            while (L > 0) {
                const int q = get8(), p = q >> 4, t = q & 15;
                if ((p != 0 && p != 1) || t > 3)
                    return false;
                for (int i = 0; i < 64; ++i)
                    dequant[t][kZigzag[i]] = (uint16_t)(p ? get16() : get8());
                L -= p ? 129 : 65;
            }
            return L == 0;
        }
        if (m == 0xC4) {
            int L = get16() - 2;
            while (L > 0) {
                const int q = get8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3)
                    return false;
                int counts[16], n = 0;
                for (int i = 0; i < 16; ++i)
                    counts[i] = get8(), n += counts[i];
                if (n > 256)
                    return false;
                uint8_t vals[256];
                for (int i = 0; i < n; ++i)
                    vals[i] = (uint8_t)get8();
                if (!(tc == 0 ? dcTables[th] : acTables[th]).build(counts, vals, n))
                    return false;
                L -= 17 + n;
            }
            return L == 0;
        }
        if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
            int L = get16();
            if (L < 2)
                return false;
            L -= 2;
            if (m == 0xE0 && L >= 5) {
                static const char tag[5] = { 'J', 'F', 'I', 'F', 0 };
                bool ok = true;
                for (int i = 0; i < 5; ++i)
                    ok = (get8() == (uint8_t)tag[i]) && ok;
                L -= 5;
                if (ok)
                    jfif = true;
            } else if (m == 0xEE && L >= 12) {
                static const char tag[6] = { 'A', 'd', 'o', 'b', 'e', 0 };
                bool ok = true;
                for (int i = 0; i < 6; ++i)
                    ok = (get8() == (uint8_t)tag[i]) && ok;
                L -= 6;
                if (ok) {
                    get8(), get16(), get16();
                    adobeTransform = get8();
                    L -= 6;
                }
            }
            pos += (size_t)L;
            return true;
        }
        return false; /* arithmetic coding, lossless and hierarchical frames, stray markers */
    }

    bool frameHeader()
    {
        const int Lf = get16();
        if (Lf < 11 || get8() != 8)
            return false;
        height = get16(), width = get16();
        if (height == 0 || width == 0)
            return false;
        /* every 8 x 8 block costs the entropy-coded data at least two bits, so a frame far larger than its file can
         * hold is refused before its planes are allocated */
        if ((uint64_t)width * height > 1024ull * size + 65536ull)
            return false;
        nComp = get8();
        if ((nComp != 1 && nComp != 3 && nComp != 4) || Lf != 8 + 3 * nComp)
            return false;
        static const char rgb[3] = { 'R', 'G', 'B' };
        for (int i = 0; i < nComp; ++i) {
            Component& c = comp[i];
            c.id = get8();
            if (nComp == 3 && c.id == rgb[i])
                ++rgbIds;
            const int q = get8();
            c.h = q >> 4, c.v = q & 15, c.tq = get8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3)
                return false;
            hMax = c.h > hMax ? c.h : hMax, vMax = c.v > vMax ? c.v : vMax;
        }
        for (int i = 0; i < nComp; ++i)
            if (hMax % comp[i].h || vMax % comp[i].v)
                return false;
        if ((uint64_t)width * height > (1ull << 30))
            return false;
        mcuX = (width + hMax * 8 - 1) / (hMax * 8), mcuY = (height + vMax * 8 - 1) / (vMax * 8);
        for (int i = 0; i < nComp; ++i) {
            Component& c = comp[i];
            c.x = (width * c.h + hMax - 1) / hMax, c.y = (height * c.v + vMax - 1) / vMax;
            c.w2 = mcuX * c.h * 8, c.h2 = mcuY * c.v * 8;
            c.pixels.assign((size_t)c.w2 * c.h2, 0);
            if (progressive)
                c.coeff.assign((size_t)c.w2 * c.h2, 0);
        }
        return true;
    }

    bool scanHeader()
    {
        const int Ls = get16();
        scanN = get8();
        if (scanN < 1 || scanN > 4 || scanN > nComp || Ls != 6 + 2 * scanN)
            return false;
        for (int i = 0; i < scanN; ++i) {
            const int id = get8(), q = get8();
            int which = 0;
            while (which < nComp && comp[which].id != id)
                ++which;
            if (which == nComp)
                return false;
            comp[which].hd = q >> 4, comp[which].ha = q & 15;
            if (comp[which].hd > 3 || comp[which].ha > 3)
                return false;
            order[i] = which;
        }
        specStart = get8(), specEnd = get8();
        const int aa = get8();
        succHigh = aa >> 4, succLow = aa & 15;
        if (progressive) {
            if (specStart > 63 || specEnd > 63 || specStart > specEnd || succHigh > 13 || succLow > 13)
                return false;
        } else {
            if (specStart != 0 || succHigh != 0 || succLow != 0)
                return false;
            specEnd = 63;
        }
        return true;
    }

    bool blockBaseline(BitReader& br, int16_t block[64], Component& c)
    {
        const Huffman &hdc = dcTables[c.hd], &hac = acTables[c.ha];
        const uint16_t* dq = dequant[c.tq];
        const int t = decodeSymbol(br, hdc);
        if (t < 0 || t > 15)
            return false;
        std::memset(block, 0, 64 * sizeof(int16_t));
        c.dcPred += receiveExtend(br, t);
        block[0] = (int16_t)(c.dcPred * dq[0]);
        int k = 1;
        do {
            const int rs = decodeSymbol(br, hac);
            if (rs < 0)
                return false;
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xF0)
                    break;
                k += 16;
            } else {
                k += r;
                const int zig = kZigzag[k++];
                block[zig] = (int16_t)(receiveExtend(br, s) * dq[zig]);
            }
        } while (k < 64);
        return true;
    }

    bool blockProgressiveDc(BitReader& br, int16_t* block, Component& c)
    {
        if (specEnd != 0)
            return false;
        if (succHigh == 0) {
            std::memset(block, 0, 64 * sizeof(int16_t));
            const int t = decodeSymbol(br, dcTables[c.hd]);
            if (t < 0 || t > 15)
                return false;
            c.dcPred += receiveExtend(br, t);
            block[0] = (int16_t)(c.dcPred * (1 << succLow));
        } else if (br.get(1)) {
            block[0] = (int16_t)(block[0] + (int16_t)(1 << succLow));
        }
        return true;
    }

    static void refine(BitReader& br, int16_t& p, int16_t bit)
    {
        if (br.get(1) && (p & bit) == 0)
            p = (int16_t)(p > 0 ? p + bit : p - bit);
    }

    bool blockProgressiveAc(BitReader& br, int16_t* block, Component& c)
    {
        if (specStart == 0)
            return false;
        const Huffman& hac = acTables[c.ha];
        if (succHigh == 0) {
            if (eobRun) {
                --eobRun;
                return true;
            }
            int k = specStart;
            do {
                const int rs = decodeSymbol(br, hac);
                if (rs < 0)
                    return false;
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eobRun = (1 << r) + (r ? br.get(r) : 0) - 1;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    block[kZigzag[k++]] = (int16_t)(receiveExtend(br, s) * (1 << succLow));
                }
            } while (k <= specEnd);
            return true;
        }
        const int16_t bit = (int16_t)(1 << succLow);
        if (eobRun) {
            --eobRun;
            for (int k = specStart; k <= specEnd; ++k) {
                int16_t& p = block[kZigzag[k]];
                if (p != 0)
                    refine(br, p, bit);
            }
            return true;
        }
        int k = specStart;
        do {
            const int rs = decodeSymbol(br, hac);
            if (rs < 0)
                return false;
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) {
                    eobRun = (1 << r) - 1 + (r ? br.get(r) : 0);
                    r = 64; /* the rest of the band only gets correction bits */
                }
            } else {
                if (s != 1)
                    return false;
                s = br.get(1) ? bit : -bit;
            }
            while (k <= specEnd) {
                int16_t& p = block[kZigzag[k++]];
                if (p != 0) {
                    refine(br, p, bit);
                } else {
                    if (r == 0) {
                        p = (int16_t)s;
                        break;
                    }
                    --r;
                }
            }
        } while (k <= specEnd);
        return true;
    }

    /* One scan's entropy-coded data.  Returns false on a damaged stream. */
    bool scanData()
    {
        BitReader br { data, size, pos };
        int todo = restartInterval ? restartInterval : 0x7fffffff;
        eobRun = 0;
        for (int i = 0; i < 4; ++i)
            comp[i].dcPred = 0;
        bool ended = false;
        /* after every MCU: at the end of a restart interval the data must be followed by an RSTn marker; anything
         * else ends the scan there (what was decoded so far is kept) */
        auto mcuDone = [&]() {
            if (--todo > 0)
                return;
            if (br.bits < 24)
                br.refill();
            if (br.marker >= 0xD0 && br.marker <= 0xD7) {
                br.reset();
                for (int i = 0; i < 4; ++i)
                    comp[i].dcPred = 0;
                eobRun = 0;
                todo = restartInterval ? restartInterval : 0x7fffffff;
            } else {
                ended = true;
            }
        };
        int16_t block[64];
        if (scanN == 1) {
            Component& c = comp[order[0]];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh && !ended; ++j)
                for (int i = 0; i < bw && !ended; ++i) {
                    if (!progressive) {
                        if (!blockBaseline(br, block, c))
                            return false;
                        inverseDct(&c.pixels[(size_t)c.w2 * j * 8 + (size_t)i * 8], c.w2, block);
                    } else {
                        int16_t* b = &c.coeff[64 * ((size_t)i + (size_t)j * (c.w2 / 8))];
                        if (!(specStart == 0 ? blockProgressiveDc(br, b, c) : blockProgressiveAc(br, b, c)))
                            return false;
                    }
                    mcuDone();
                }
        } else {
            for (int j = 0; j < mcuY && !ended; ++j)
                for (int i = 0; i < mcuX && !ended; ++i) {
                    for (int k = 0; k < scanN; ++k) {
                        Component& c = comp[order[k]];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x) {
                                const int bx = i * c.h + x, by = j * c.v + y;
                                if (!progressive) {
                                    if (!blockBaseline(br, block, c))
                                        return false;
                                    inverseDct(&c.pixels[(size_t)c.w2 * by * 8 + (size_t)bx * 8], c.w2, block);
                                } else if (!blockProgressiveDc(br, &c.coeff[64 * ((size_t)bx + (size_t)by * (c.w2 / 8))], c)) {
                                    return false; /* interleaved progressive scans carry DC coefficients only */
                                }
                            }
                    }
                    mcuDone();
                }
        }
        pos = br.pos;
        pendingMarker = br.marker;
        if (pendingMarker < 0) {
            /* no marker met while reading: skip to the next 0xFF (some writers pad the data with zeros) */
            while (!atEnd()) {
                if (get8() == 255) {
                    pendingMarker = get8();
                    break;
                }
            }
        }
        return true;
    }

    void finishProgressive()
    {
        for (int n = 0; n < nComp; ++n) {
            Component& c = comp[n];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i) {
                    int16_t* b = &c.coeff[64 * ((size_t)i + (size_t)j * (c.w2 / 8))];
                    for (int k = 0; k < 64; ++k)
                        b[k] = (int16_t)(b[k] * dequant[c.tq][k]);
                    inverseDct(&c.pixels[(size_t)c.w2 * j * 8 + (size_t)i * 8], c.w2, b);
                }
        }
    }

    bool decode()
    {
        if (nextMarker() != 0xD8)
            return false;
        int m = nextMarker();
        while (!(m == 0xC0 || m == 0xC1 || m == 0xC2)) {
            if (!tablesAndSegments(m))
                return false;
            m = nextMarker();
            while (m < 0) { /* padding between segments */
                if (atEnd())
                    return false;
                m = nextMarker();
            }
        }
        progressive = m == 0xC2;
        if (!frameHeader())
            return false;
        m = nextMarker();
        while (m != 0xD9) {
            if (m == 0xDA) {
                if (!scanHeader() || !scanData())
                    return false;
            } else if (m == 0xDC) {
                const int Ld = get16(), NL = get16();
                if (Ld != 4 || NL != height)
                    return false;
            } else if (!tablesAndSegments(m)) {
                return false;
            }
            m = nextMarker();
        }
        if (progressive)
            finishProgressive();
        return true;
    }
};

/* chroma rows to full width: the four upsampling kernels */
const uint8_t* upsampleRow(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int hs, int vs)
{
    if (hs == 1 && vs == 1)
        return near;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; ++i)
            out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
        return out;
    }
    if (hs == 2 && vs == 1) {
        if (w == 1) {
            out[0] = out[1] = near[0];
            return out;
        }
        out[0] = near[0];
        out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
        int i = 1;
        for (; i < w - 1; ++i) {
            const int n = 3 * near[i] + 2;
            out[i * 2] = (uint8_t)((n + near[i - 1]) >> 2);
            out[i * 2 + 1] = (uint8_t)((n + near[i + 1]) >> 2);
        }
        out[i * 2] = (uint8_t)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
        out[i * 2 + 1] = near[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) {
            out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2);
            return out;
        }
        int t1 = 3 * near[0] + far[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; ++i) {
            const int t0 = t1;
            t1 = 3 * near[i] + far[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < hs; ++j)
            out[i * hs + j] = near[i];
    return out;
}

inline uint8_t mul8(uint8_t x, uint8_t y)
{
    const unsigned t = (unsigned)x * y + 128;
    return (uint8_t)((t + (t >> 8)) >> 8);
}

void ycbcrRow(uint8_t* out, const uint8_t* y, const uint8_t* pcb, const uint8_t* pcr, int count)
{
    /* ((int)(c * 4096.0f + 0.5f)) << 8 for 1.40200, 0.71414, 0.34414, 1.77200 */
    const int crR = 5743 << 8, crG = 2925 << 8, cbG = 1410 << 8, cbB = 7258 << 8;
    for (int i = 0; i < count; ++i) {
        const int yf = (y[i] << 20) + (1 << 19);
        const int cr = pcr[i] - 128, cb = pcb[i] - 128;
        int r = yf + cr * crR;
        int g = yf + cr * -crG + (int)((uint32_t)(cb * -cbG) & 0xffff0000u);
        int b = yf + cb * cbB;
        r >>= 20, g >>= 20, b >>= 20;
        out[3 * i] = clampByte(r), out[3 * i + 1] = clampByte(g), out[3 * i + 2] = clampByte(b);
    }
}

} // namespace

/* JPEG -> RGB8, what stb_image returns for three requested channels.  False when the file is not a JPEG this
 * decoder (or stb_image) reads. */
bool decodeJpeg(const std::vector<uint8_t>& file, int& w, int& h, std::vector<uint8_t>& rgb)
{
    if (file.size() < 4 || file[0] != 0xFF || file[1] != 0xD8)
        return false;
    Decoder d { file.data(), file.size() };
    if (!d.decode())
        return false;
    w = d.width, h = d.height;
    rgb.assign((size_t)w * h * 3, 0);
    const bool isRgb = d.nComp == 3 && (d.rgbIds == 3 || (d.adobeTransform == 0 && !d.jfif));
    struct Resample {
        int hs, vs, ystep, ypos, wLores;
        const uint8_t *line0, *line1;
        std::vector<uint8_t> buffer;
    } rs[4];
    for (int k = 0; k < d.nComp; ++k) {
        Resample& r = rs[k];
        r.hs = d.hMax / d.comp[k].h, r.vs = d.vMax / d.comp[k].v;
        r.ystep = r.vs >> 1, r.ypos = 0;
        r.wLores = (w + r.hs - 1) / r.hs;
        r.line0 = r.line1 = d.comp[k].pixels.data();
        r.buffer.assign((size_t)w + 8, 0);
    }
    const uint8_t* rows[4] = { nullptr, nullptr, nullptr, nullptr };
    for (int j = 0; j < h; ++j) {
        uint8_t* out = &rgb[(size_t)j * w * 3];
        for (int k = 0; k < d.nComp; ++k) {
            Resample& r = rs[k];
            /* the row nearer to this output row, and the other one of the pair it lies between */
            const bool bottom = r.ystep >= (r.vs >> 1);
            rows[k] = upsampleRow(r.buffer.data(), bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.wLores, r.hs, r.vs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < d.comp[k].y)
                    r.line1 += d.comp[k].w2;
            }
        }
        if (d.nComp == 3) {
            if (isRgb)
                for (int i = 0; i < w; ++i)
                    out[3 * i] = rows[0][i], out[3 * i + 1] = rows[1][i], out[3 * i + 2] = rows[2][i];
            else
                ycbcrRow(out, rows[0], rows[1], rows[2], w);
        } else if (d.nComp == 4) {
            if (d.adobeTransform == 0) {
                for (int i = 0; i < w; ++i) {
                    const uint8_t m = rows[3][i];
                    out[3 * i] = mul8(rows[0][i], m), out[3 * i + 1] = mul8(rows[1][i], m), out[3 * i + 2] = mul8(rows[2][i], m);
                }
            } else {
                ycbcrRow(out, rows[0], rows[1], rows[2], w);
                if (d.adobeTransform == 2)
                    for (int i = 0; i < w; ++i) {
                        const uint8_t m = rows[3][i];
                        out[3 * i] = mul8((uint8_t)(255 - out[3 * i]), m), out[3 * i + 1] = mul8((uint8_t)(255 - out[3 * i + 1]), m);
                        out[3 * i + 2] = mul8((uint8_t)(255 - out[3 * i + 2]), m);
                    }
            }
        } else {
            for (int i = 0; i < w; ++i)
                out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = rows[0][i];
        }
    }
    return true;
}

} // namespace srt
