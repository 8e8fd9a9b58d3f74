// gltf_loader.cpp -- the scene-loader stand-in behind include/arctic_gltf.h.
//
// Restates what App::load_scene (reference src/app.cpp:173-385) obtains from assimp + stb_image for a glTF 2.0 file, with
// nothing but the C++ standard library and zlib: a small JSON reader, a PNG decoder, the glTF accessor plumbing, and the
// post-processing steps the reference asks assimp for (Triangulate is moot: triangles only; FlipUVs; CalcTangentSpace;
// JoinIdenticalVertices is skipped -- glTF vertices are already indexed and joining does not change the image).
// Host code only: no HIP here.  assimp and the real assets are not available offline, so this file is checked against
// glTF files the test-suite writes itself (tests/test_gltf_loader.py); parity with assimp's output is unpinned.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/arctic_gltf.h"

namespace {

[[noreturn]] void fail(const std::string &m) { throw std::runtime_error(m); }

// ------------------------------------------------------------------------------------------------- JSON
struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json *find(const char *key) const {
        if (kind != Object) return nullptr;
        for (const auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Json &at(const char *key) const {
        const Json *j = find(key);
        if (!j) fail(std::string("glTF: missing key '") + key + "'");
        return *j;
    }
    bool has(const char *key) const { return find(key) != nullptr; }
    size_t size() const { return kind == Array ? arr.size() : 0; }
    const Json &operator[](size_t i) const {
        if (kind != Array || i >= arr.size()) fail("glTF: array index out of range");
        return arr[i];
    }
    int64_t as_int() const {
        if (kind != Number) fail("glTF: number expected");
        return (int64_t)num;
    }
    double as_num() const {
        if (kind != Number) fail("glTF: number expected");
        return num;
    }
    const std::string &as_str() const {
        if (kind != String) fail("glTF: string expected");
        return str;
    }
};

struct JsonParser {
    const char *p, *end;
    explicit JsonParser(const std::string &s) : p(s.data()), end(s.data() + s.size()) {}
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool eat(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
    void expect(char c) { if (!eat(c)) fail(std::string("JSON: expected '") + c + "'"); }
    Json parse() { Json j = value(); ws(); if (p != end) fail("JSON: trailing characters"); return j; }
    Json value() {
        ws();
        if (p >= end) fail("JSON: unexpected end");
        Json j;
        if (*p == '{') {
            ++p; j.kind = Json::Object;
            if (eat('}')) return j;
            do { ws(); std::string k = string(); expect(':'); j.obj.emplace_back(std::move(k), value()); } while (eat(','));
            expect('}');
        } else if (*p == '[') {
            ++p; j.kind = Json::Array;
            if (eat(']')) return j;
            do { j.arr.push_back(value()); } while (eat(','));
            expect(']');
        } else if (*p == '"') {
            j.kind = Json::String; j.str = string();
        } else if (!std::strncmp(p, "true", 4)) { j.kind = Json::Bool; j.b = true; p += 4;
        } else if (!std::strncmp(p, "false", 5)) { j.kind = Json::Bool; p += 5;
        } else if (!std::strncmp(p, "null", 4)) { p += 4;
        } else {
            char *e = nullptr;
            j.kind = Json::Number; j.num = std::strtod(p, &e);
            if (e == p) fail("JSON: bad value");
            p = e;
        }
        return j;
    }
    std::string string() {
        if (p >= end || *p != '"') fail("JSON: string expected");
        ++p;
        std::string s;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) break;
                switch (*p) {
                case 'n': s += '\n'; break; case 't': s += '\t'; break; case 'r': s += '\r'; break;
                case 'b': s += '\b'; break; case 'f': s += '\f'; break;
                case 'u': {   // BMP code point -> UTF-8 (surrogate pairs are not needed for file names)
                    if (end - p < 5) fail("JSON: bad \\u escape");
                    unsigned cp = (unsigned)std::strtoul(std::string(p + 1, 4).c_str(), nullptr, 16);
                    p += 4;
                    if (cp < 0x80) s += (char)cp;
                    else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
                    else { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: s += *p;
                }
                ++p;
            } else s += *p++;
        }
        if (p >= end) fail("JSON: unterminated string");
        ++p;
        return s;
    }
};

// ------------------------------------------------------------------------------------------------- PNG
uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// RGBA8 like stbi_load(path, &w, &h, nullptr, 4): grey is replicated, missing alpha is 255, 16-bit keeps the high byte
std::vector<uint8_t> decode_png(const uint8_t *data, size_t size, uint32_t &w, uint32_t &h) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (size < 8 || std::memcmp(data, sig, 8)) fail("not a PNG or baseline JPEG file");
    size_t pos = 8;
    uint32_t depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool have_ihdr = false;
    while (pos + 12 <= size) {
        const uint32_t len = be32(data + pos);
        const char *type = reinterpret_cast<const char *>(data + pos + 4);
        const uint8_t *body = data + pos + 8;
        if (pos + 12 + (size_t)len > size) fail("PNG: truncated chunk");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) fail("PNG: bad IHDR");
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w == 0 || h == 0 || w > 65535 || h > 65535) fail("PNG: bad header");
    if (interlace) fail("PNG: interlaced images are not supported");
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels) fail("PNG: bad colour type");
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) fail("PNG: unsupported bit depth");
    if (ctype == 3 && plte.size() < 3) fail("PNG: palette missing");
    const size_t bpp_bits = (size_t)channels * depth, stride = (w * bpp_bits + 7) / 8, bpp = (bpp_bits + 7) / 8;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf out_len = (uLongf)raw.size();
    int zr = uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || out_len != raw.size()) fail("PNG: inflate failed");
    std::vector<uint8_t> prev(stride, 0), line(stride);
    std::vector<uint8_t> out((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t *src = &raw[(stride + 1) * y + 1];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? line[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int v = src[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: fail("PNG: bad filter type");
            }
            line[i] = (uint8_t)v;
        }
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t *o = &out[((size_t)y * w + x) * 4];
            auto sample = [&](int ch) -> uint32_t {   // channel ch of pixel x, scaled to 8 bits
                if (depth == 8) return line[(size_t)x * channels + ch];
                if (depth == 16) return line[((size_t)x * channels + ch) * 2];
                const size_t bit = (size_t)x * depth;
                const uint32_t v = (line[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                return ctype == 3 ? v : v * 255u / ((1u << depth) - 1u);
            };
            switch (ctype) {
            case 0: { const uint32_t g = sample(0); o[0] = o[1] = o[2] = (uint8_t)g; o[3] = 255; break; }
            case 2: o[0] = (uint8_t)sample(0); o[1] = (uint8_t)sample(1); o[2] = (uint8_t)sample(2); o[3] = 255; break;
            case 3: {
                const uint32_t i = sample(0);
                if ((size_t)i * 3 + 2 >= plte.size()) fail("PNG: palette index out of range");
                o[0] = plte[i * 3]; o[1] = plte[i * 3 + 1]; o[2] = plte[i * 3 + 2]; o[3] = i < trns.size() ? trns[i] : 255;
                break;
            }
            case 4: { const uint32_t g = sample(0); o[0] = o[1] = o[2] = (uint8_t)g; o[3] = (uint8_t)sample(1); break; }
            default: o[0] = (uint8_t)sample(0); o[1] = (uint8_t)sample(1); o[2] = (uint8_t)sample(2); o[3] = (uint8_t)sample(3);
            }
        }
        prev.swap(line);
    }
    return out;
}

// ------------------------------------------------------------------------------------------------- JPEG (baseline)
// Baseline sequential DCT, Huffman, 8-bit, 1 or 3 components, any sampling factors up to 2x2, restart intervals -- what the
// Khronos sample assets use.  Float IDCT, chroma upsampled by pixel replication, JFIF YCbCr -> RGB: stb_image (which the
// reference uses) has its own integer IDCT and a smoothing upsampler, so results can differ from it by an LSB or two.
struct JpegBits {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    bool hit_marker = false;
    void fill() {
        while (n <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    const int c = p < end ? *p : 0xD9;
                    if (c == 0) ++p;                      // stuffed zero
                    else { hit_marker = true; --p; b = 0; }   // a marker: feed zeros until the caller deals with it
                }
            }
            acc |= (uint32_t)b << (24 - n);
            n += 8;
        }
    }
    int bits(int k) {
        if (k == 0) return 0;
        if (n < k) fill();
        const int v = (int)(acc >> (32 - k));
        acc <<= k; n -= k;
        return v;
    }
    void reset() { acc = 0; n = 0; hit_marker = false; }
};
struct JpegHuff {
    uint8_t count[17] = {0}, symbol[256] = {0};
    int mincode[17] = {0}, maxcode[18] = {0}, valptr[17] = {0};
    void build() {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k; mincode[len] = code;
            code += count[len]; k += count[len];
            maxcode[len] = count[len] ? code - 1 : -1;
            code <<= 1;
        }
    }
    int decode(JpegBits &br) const {
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = (code << 1) | br.bits(1);
            if (maxcode[len] >= 0 && code <= maxcode[len] && code >= mincode[len]) return symbol[valptr[len] + code - mincode[len]];
        }
        fail("JPEG: bad Huffman code");
    }
};
inline int jpeg_extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

std::vector<uint8_t> decode_jpeg(const uint8_t *data, size_t size, uint32_t &w, uint32_t &h) {
    static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) fail("not a JPEG file");
    uint16_t qt[4][64] = {};
    JpegHuff dc[4], ac[4];
    struct Comp { int id = 0, hs = 1, vs = 1, tq = 0, td = 0, ta = 0, pred = 0, bw = 0, bh = 0; std::vector<uint8_t> plane; } comp[3];
    int ncomp = 0, restart = 0, hmax = 1, vmax = 1;
    bool have_sof = false, have_dc[4] = {false, false, false, false}, have_ac[4] = {false, false, false, false};
    size_t pos = 2;
    auto be16 = [&](size_t o) { if (o + 2 > size) fail("JPEG: truncated"); return (int)((data[o] << 8) | data[o + 1]); };
    for (;;) {
        if (pos + 4 > size) fail("JPEG: no scan found");
        if (data[pos] != 0xFF) { ++pos; continue; }
        const int m = data[pos + 1];
        if (m == 0xFF) { ++pos; continue; }
        pos += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        const int len = be16(pos);
        if (len < 2 || pos + (size_t)len > size) fail("JPEG: bad segment length");
        const uint8_t *seg = data + pos + 2;
        const int n = len - 2;
        if (m == 0xDB) {   // DQT
            for (int o = 0; o < n;) {
                const int pq = seg[o] >> 4, tq = seg[o] & 15; ++o;
                if (tq > 3 || pq > 1) fail("JPEG: bad quantisation table id");
                if (o + (pq ? 128 : 64) > n) fail("JPEG: truncated quantisation table");
                for (int i = 0; i < 64; ++i) { qt[tq][zz[i]] = pq ? (uint16_t)((seg[o] << 8) | seg[o + 1]) : seg[o]; o += pq ? 2 : 1; }
            }
        } else if (m == 0xC4) {   // DHT
            for (int o = 0; o < n;) {
                const int tc = seg[o] >> 4, th = seg[o] & 15; ++o;
                if (th > 3 || tc > 1) fail("JPEG: bad Huffman table id");
                if (o + 16 > n) fail("JPEG: truncated Huffman table");
                JpegHuff &t = tc ? ac[th] : dc[th];
                (tc ? have_ac : have_dc)[th] = true;
                int total = 0;
                for (int i = 1; i <= 16; ++i) { t.count[i] = seg[o++]; total += t.count[i]; }
                if (total > 256 || o + total > n) fail("JPEG: bad Huffman table");
                std::memcpy(t.symbol, seg + o, (size_t)total); o += total;
                t.build();
            }
        } else if (m == 0xC0 || m == 0xC1) {   // SOF0 / SOF1 (sequential Huffman)
            if (n < 6) fail("JPEG: truncated frame header");
            if (seg[0] != 8) fail("JPEG: only 8-bit precision is supported");
            h = (uint32_t)((seg[1] << 8) | seg[2]); w = (uint32_t)((seg[3] << 8) | seg[4]); ncomp = seg[5];
            if ((ncomp != 1 && ncomp != 3) || !w || !h) fail("JPEG: unsupported component count or size");
            if (n < 6 + 3 * ncomp) fail("JPEG: truncated frame header");
            for (int c = 0; c < ncomp; ++c) {
                comp[c].id = seg[6 + c * 3]; comp[c].hs = seg[7 + c * 3] >> 4; comp[c].vs = seg[7 + c * 3] & 15; comp[c].tq = seg[8 + c * 3];
                if (comp[c].hs < 1 || comp[c].hs > 2 || comp[c].vs < 1 || comp[c].vs > 2 || comp[c].tq > 3) fail("JPEG: unsupported sampling factors");
                hmax = std::max(hmax, comp[c].hs); vmax = std::max(vmax, comp[c].vs);
            }
            have_sof = true;
        } else if (m == 0xC2) fail("JPEG: progressive files are not supported (re-save as baseline or PNG)");
        else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) fail("JPEG: unsupported coding process");
        else if (m == 0xDD) { if (n < 2) fail("JPEG: truncated restart interval"); restart = (seg[0] << 8) | seg[1]; }
        else if (m == 0xDA) {   // SOS: the one interleaved scan of a baseline file
            if (!have_sof) fail("JPEG: scan before frame header");
            if (n < 1 || seg[0] != ncomp) fail("JPEG: non-interleaved scans are not supported");
            if (n < 1 + 2 * ncomp + 3) fail("JPEG: truncated scan header");
            for (int i = 0; i < ncomp; ++i) {
                const int id = seg[1 + i * 2];
                int c = 0;
                while (c < ncomp && comp[c].id != id) ++c;
                if (c == ncomp) fail("JPEG: scan refers to an unknown component");
                comp[c].td = seg[2 + i * 2] >> 4; comp[c].ta = seg[2 + i * 2] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3 || !have_dc[comp[c].td] || !have_ac[comp[c].ta]) fail("JPEG: scan selects a Huffman table that was never defined");
            }
            pos += (size_t)len;
            break;
        }
        pos += (size_t)len;
    }
    const int mcux = (int)((w + 8u * hmax - 1) / (8u * hmax)), mcuy = (int)((h + 8u * vmax - 1) / (8u * vmax));
    for (int c = 0; c < ncomp; ++c) {
        comp[c].bw = mcux * comp[c].hs * 8; comp[c].bh = mcuy * comp[c].vs * 8;
        comp[c].plane.assign((size_t)comp[c].bw * comp[c].bh, 0);
    }
    // cos table of the 8-point IDCT
    float ct[8][8];
    for (int x = 0; x < 8; ++x) for (int u = 0; u < 8; ++u) ct[x][u] = (u ? 1.0f : 0.70710678f) * 0.5f * std::cos((2 * x + 1) * u * 3.14159265358979f / 16.0f);
    JpegBits br{data + pos, data + size};
    int until_restart = restart;
    for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
            if (restart && until_restart == 0) {   // RSTn: byte-align, skip the marker, reset the predictors
                const uint8_t *q = br.p;
                while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                br.p = q + 2 <= br.end ? q + 2 : br.end;
                br.reset();
                for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
                until_restart = restart;
            }
            for (int c = 0; c < ncomp; ++c)
                for (int by = 0; by < comp[c].vs; ++by)
                    for (int bx = 0; bx < comp[c].hs; ++bx) {
                        float blk[64] = {0};
                        const int t = dc[comp[c].td].decode(br);
                        if (t > 11) fail("JPEG: bad DC category");
                        comp[c].pred += jpeg_extend(br.bits(t), t);
                        blk[0] = (float)(comp[c].pred * qt[comp[c].tq][0]);
                        for (int k = 1; k < 64;) {
                            const int rs = ac[comp[c].ta].decode(br), r = rs >> 4, sz = rs & 15;
                            if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                            if (sz > 10) fail("JPEG: bad AC coefficient size");
                            k += r;
                            if (k > 63) fail("JPEG: coefficient index out of range");
                            blk[zz[k]] = (float)(jpeg_extend(br.bits(sz), sz) * qt[comp[c].tq][zz[k]]);
                            ++k;
                        }
                        float tmp[64];
                        for (int y = 0; y < 8; ++y)
                            for (int x = 0; x < 8; ++x) { float a = 0; for (int u = 0; u < 8; ++u) a += ct[x][u] * blk[y * 8 + u]; tmp[y * 8 + x] = a; }
                        uint8_t *dst = &comp[c].plane[(size_t)((my * comp[c].vs + by) * 8) * comp[c].bw + (size_t)(mx * comp[c].hs + bx) * 8];
                        for (int x = 0; x < 8; ++x)
                            for (int y = 0; y < 8; ++y) {
                                float a = 0;
                                for (int v = 0; v < 8; ++v) a += ct[y][v] * tmp[v * 8 + x];
                                const int px = (int)std::lrint(a + 128.0f);
                                dst[(size_t)y * comp[c].bw + x] = (uint8_t)(px < 0 ? 0 : px > 255 ? 255 : px);
                            }
                    }
            if (restart) --until_restart;
        }
    std::vector<uint8_t> out((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            auto at = [&](int c) { return (int)comp[c].plane[(size_t)(y * comp[c].vs / vmax) * comp[c].bw + x * comp[c].hs / hmax]; };
            uint8_t *o = &out[((size_t)y * w + x) * 4];
            if (ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)at(0); }
            else {
                const float Y = (float)at(0), cb = (float)at(1) - 128.0f, cr = (float)at(2) - 128.0f;
                const float rgb[3] = {Y + 1.402f * cr, Y - 0.344136f * cb - 0.714136f * cr, Y + 1.772f * cb};
                for (int k = 0; k < 3; ++k) { const int v = (int)std::lrint(rgb[k]); o[k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
            }
            o[3] = 255;
        }
    return out;
}

// either format, by signature (stbi_load does the same)
std::vector<uint8_t> decode_image(const uint8_t *data, size_t size, uint32_t &w, uint32_t &h) {
    if (size >= 2 && data[0] == 0xFF && data[1] == 0xD8) return decode_jpeg(data, size, w, h);
    return decode_png(data, size, w, h);
}

// ------------------------------------------------------------------------------------------------- files, base64
std::vector<uint8_t> read_file(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("cannot open '" + path + "'");
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

std::vector<uint8_t> base64(const std::string &s, size_t from) {
    std::vector<uint8_t> out;
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = from; i < s.size(); ++i) {
        const char c = s[i];
        int v = c >= 'A' && c <= 'Z' ? c - 'A' : c >= 'a' && c <= 'z' ? c - 'a' + 26 : c >= '0' && c <= '9' ? c - '0' + 52 : c == '+' ? 62 : c == '/' ? 63 : -1;
        if (v < 0) continue;   // '=' padding, whitespace
        acc = (acc << 6) | (uint32_t)v;
        if ((bits += 6) >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
    }
    return out;
}

std::string dir_of(const std::string &path) {
    const size_t s = path.find_last_of("/\\");
    return s == std::string::npos ? std::string() : path.substr(0, s + 1);
}

std::string uri_decode(const std::string &u) {
    std::string o;
    for (size_t i = 0; i < u.size(); ++i)
        if (u[i] == '%' && i + 2 < u.size()) { o += (char)std::strtoul(u.substr(i + 1, 2).c_str(), nullptr, 16); i += 2; }
        else o += u[i];
    return o;
}

std::vector<uint8_t> load_uri(const std::string &uri, const std::string &base) {
    if (uri.rfind("data:", 0) == 0) {
        const size_t c = uri.find("base64,");
        if (c == std::string::npos) fail("glTF: only base64 data URIs are supported");
        return base64(uri, c + 7);
    }
    return read_file(base + uri_decode(uri));
}

// ------------------------------------------------------------------------------------------------- math (column-major 4x4 as glm)
struct M4 { float m[16]; };   // m[col * 4 + row]
M4 identity() { M4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
M4 mul(const M4 &a, const M4 &b) {   // glm operator*: column j of the result = a * (column j of b), fp32 left to right
    M4 r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r.m[j * 4 + i] = ((a.m[i] * b.m[j * 4] + a.m[4 + i] * b.m[j * 4 + 1]) + a.m[8 + i] * b.m[j * 4 + 2]) + a.m[12 + i] * b.m[j * 4 + 3];
    return r;
}
M4 transpose(const M4 &a) { M4 r; for (int c = 0; c < 4; ++c) for (int q = 0; q < 4; ++q) r.m[c * 4 + q] = a.m[q * 4 + c]; return r; }

struct Image { std::vector<uint8_t> px; uint32_t w = 0, h = 0; };
Image solid(uint8_t r, uint8_t g, uint8_t b) {   // assets/white.png, assets/normal.png are 16x16 of one colour
    Image im; im.w = im.h = 16; im.px.resize(16 * 16 * 4);
    for (size_t i = 0; i < 256; ++i) { im.px[i * 4] = r; im.px[i * 4 + 1] = g; im.px[i * 4 + 2] = b; im.px[i * 4 + 3] = 255; }
    return im;
}

struct Material { Image img[3]; };
struct Mesh { std::vector<ArcticVertex> v; std::vector<uint32_t> idx; uint64_t material = 0; };

}  // namespace

struct ArcticGltf {
    std::vector<Material> materials;
    std::vector<Mesh> meshes;
    std::vector<ArcticObject> objects;
};

namespace {

struct Loader {
    const Json &doc;
    std::string base;
    std::vector<std::vector<uint8_t>> buffers;
    std::vector<uint8_t> glb_bin;   // the BIN chunk of a .glb: buffer 0 when that buffer has no uri
    std::map<size_t, Image> image_cache;

    Loader(const Json &d, std::string b) : doc(d), base(std::move(b)) {}

    // every index and offset of the file is untrusted: non-negative and in range before use, ranges checked without overflow
    static size_t index_of(const Json &j, const char *what) {
        const double v = j.as_num();
        if (!(v >= 0.0) || v > 4.0e15 || v != std::floor(v)) fail(std::string("glTF: bad ") + what);
        return (size_t)v;
    }
    static void check_range(size_t off, size_t need, size_t size, const char *what) {
        if (off > size || need > size - off) fail(std::string("glTF: ") + what + " out of bounds");
    }
    const std::vector<uint8_t> &buffer(size_t i) {
        const Json &bufs = doc.at("buffers");
        const Json &b = bufs[i];   // range-checked
        if (buffers.size() < bufs.size()) buffers.resize(bufs.size());
        if (buffers[i].empty()) {
            if (b.has("uri")) buffers[i] = load_uri(b.at("uri").as_str(), base);
            else if (i == 0 && !glb_bin.empty()) buffers[i] = glb_bin;
            else fail("glTF: buffer without uri");
            if (buffers[i].size() < index_of(b.at("byteLength"), "byteLength")) fail("glTF: buffer shorter than byteLength");
        }
        return buffers[i];
    }

    // accessor -> floats (n_comp per element), converting normalised integers like the glTF spec says
    std::vector<float> floats(size_t accessor, int n_comp) {
        const Json &a = doc.at("accessors")[accessor];
        if (a.has("sparse")) fail("glTF: sparse accessors are not supported");
        static const std::map<std::string, int> comps = {{"SCALAR", 1}, {"VEC2", 2}, {"VEC3", 3}, {"VEC4", 4}};
        const auto it = comps.find(a.at("type").as_str());
        if (it == comps.end() || it->second < n_comp) fail("glTF: accessor type mismatch");
        const int file_comp = it->second;
        const int64_t ct = a.at("componentType").as_int();
        const size_t count = index_of(a.at("count"), "accessor count");
        const bool norm = a.has("normalized") && a.at("normalized").b;
        const size_t csize = ct == 5126 || ct == 5125 ? 4 : (ct == 5123 || ct == 5122 ? 2 : 1);
        const Json &bv = doc.at("bufferViews")[index_of(a.at("bufferView"), "bufferView index")];
        const std::vector<uint8_t> &buf = buffer(index_of(bv.at("buffer"), "buffer index"));
        const size_t off_v = bv.has("byteOffset") ? index_of(bv.at("byteOffset"), "byteOffset") : 0, off_a = a.has("byteOffset") ? index_of(a.at("byteOffset"), "byteOffset") : 0;
        const size_t stride = bv.has("byteStride") ? index_of(bv.at("byteStride"), "byteStride") : csize * file_comp;
        const size_t elem = csize * file_comp;
        if (stride < elem || stride > 65536) fail("glTF: bad byteStride");
        check_range(off_v, off_a, buf.size(), "accessor");
        const size_t off = off_v + off_a;
        if (count) { if (count - 1 > (buf.size() - off) / stride) fail("glTF: accessor out of bounds"); check_range(off, (count - 1) * stride + elem, buf.size(), "accessor"); }
        std::vector<float> out(count * (size_t)n_comp);
        for (size_t i = 0; i < count; ++i)
            for (int c = 0; c < n_comp; ++c) {
                const uint8_t *p = &buf[off + (size_t)i * stride + (size_t)c * csize];
                float v;
                switch (ct) {
                case 5126: std::memcpy(&v, p, 4); break;
                case 5121: v = norm ? *p / 255.0f : (float)*p; break;
                case 5123: { uint16_t u; std::memcpy(&u, p, 2); v = norm ? u / 65535.0f : (float)u; break; }
                case 5120: { int8_t s; std::memcpy(&s, p, 1); v = norm ? std::fmax(s / 127.0f, -1.0f) : (float)s; break; }
                case 5122: { int16_t s; std::memcpy(&s, p, 2); v = norm ? std::fmax(s / 32767.0f, -1.0f) : (float)s; break; }
                default: fail("glTF: unsupported component type");
                }
                out[(size_t)i * n_comp + c] = v;
            }
        return out;
    }

    std::vector<uint32_t> indices(size_t accessor) {
        const Json &a = doc.at("accessors")[accessor];
        const int64_t ct = a.at("componentType").as_int();
        const size_t count = index_of(a.at("count"), "accessor count");
        const size_t csize = ct == 5125 ? 4 : ct == 5123 ? 2 : ct == 5121 ? 1 : 0;
        if (!csize || a.at("type").as_str() != "SCALAR") fail("glTF: bad index accessor");
        const Json &bv = doc.at("bufferViews")[index_of(a.at("bufferView"), "bufferView index")];
        const std::vector<uint8_t> &buf = buffer(index_of(bv.at("buffer"), "buffer index"));
        const size_t off_v = bv.has("byteOffset") ? index_of(bv.at("byteOffset"), "byteOffset") : 0, off_a = a.has("byteOffset") ? index_of(a.at("byteOffset"), "byteOffset") : 0;
        check_range(off_v, off_a, buf.size(), "index accessor");
        const size_t off = off_v + off_a;
        if (count > (buf.size() - off) / csize) fail("glTF: index accessor out of bounds");
        std::vector<uint32_t> out(count);
        for (size_t i = 0; i < count; ++i) {
            const uint8_t *p = &buf[off + (size_t)i * csize];
            if (csize == 4) std::memcpy(&out[(size_t)i], p, 4);
            else if (csize == 2) { uint16_t u; std::memcpy(&u, p, 2); out[(size_t)i] = u; }
            else out[(size_t)i] = *p;
        }
        return out;
    }

    const Image &image_of_texture(size_t texture) {
        const Json &t = doc.at("textures")[texture];
        const size_t src = index_of(t.at("source"), "image index");
        auto it = image_cache.find(src);
        if (it != image_cache.end()) return it->second;
        const Json &im = doc.at("images")[src];
        std::vector<uint8_t> file;
        if (im.has("uri")) file = load_uri(im.at("uri").as_str(), base);
        else if (im.has("bufferView")) {
            const Json &bv = doc.at("bufferViews")[index_of(im.at("bufferView"), "bufferView index")];
            const std::vector<uint8_t> &buf = buffer(index_of(bv.at("buffer"), "buffer index"));
            const size_t off = bv.has("byteOffset") ? index_of(bv.at("byteOffset"), "byteOffset") : 0, len = index_of(bv.at("byteLength"), "byteLength");
            check_range(off, len, buf.size(), "image bufferView");
            file.assign(buf.begin() + (long)off, buf.begin() + (long)(off + len));
        } else fail("glTF: image without uri or bufferView");
        Image out;
        out.px = decode_image(file.data(), file.size(), out.w, out.h);
        return image_cache.emplace(src, std::move(out)).first->second;
    }
};

inline void sub3(const float *a, const float *b, float *o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
inline float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void cross3(const float *a, const float *b, float *o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
inline bool normalize3(float *v) { const float l = std::sqrt(dot3(v, v)); if (!(l > 0.0f) || !std::isfinite(l)) return false; v[0] /= l; v[1] /= l; v[2] /= l; return true; }

// aiProcess_CalcTangentSpace (assimp v5.4.3, the version the reference pins in CMakeLists.txt:82-87; restated from its
// published algorithm) for meshes without a TANGENT attribute: per triangle, the UV-space gradient of the position
// (dirCorrection keeps handedness), orthogonalised against each vertex normal and normalised; a vertex keeps the value of
// the last triangle that touches it (assimp additionally averages over vertices sharing position and normal).
void calc_tangents(Mesh &m) {
    for (size_t f = 0; f + 2 < m.idx.size(); f += 3) {
        ArcticVertex *p[3] = {&m.v[m.idx[f]], &m.v[m.idx[f + 1]], &m.v[m.idx[f + 2]]};
        float v[3], w[3];
        sub3(p[1]->position, p[0]->position, v);
        sub3(p[2]->position, p[0]->position, w);
        float sx = p[1]->tex_coords[0] - p[0]->tex_coords[0], sy = p[1]->tex_coords[1] - p[0]->tex_coords[1];
        float tx = p[2]->tex_coords[0] - p[0]->tex_coords[0], ty = p[2]->tex_coords[1] - p[0]->tex_coords[1];
        const float dir = (tx * sy - ty * sx) < 0.0f ? -1.0f : 1.0f;
        if (sx * ty == sy * tx) { sx = 0.0f; sy = 1.0f; tx = 1.0f; ty = 0.0f; }
        float t[3], b[3];
        for (int k = 0; k < 3; ++k) { t[k] = (w[k] * sy - v[k] * ty) * dir; b[k] = (-w[k] * sx + v[k] * tx) * dir; }   // +dP/du, +dP/dv
        for (ArcticVertex *q : p) {
            float lt[3], lb[3];
            const float dt = dot3(t, q->normal), db = dot3(b, q->normal);
            for (int k = 0; k < 3; ++k) { lt[k] = t[k] - q->normal[k] * dt; lb[k] = b[k] - q->normal[k] * db; }
            const bool ok_t = normalize3(lt), ok_b = normalize3(lb);
            if (!ok_t && ok_b) { cross3(lb, q->normal, lt); normalize3(lt); }
            if (!ok_b && ok_t) { cross3(q->normal, lt, lb); normalize3(lb); }
            std::memcpy(q->tangent, lt, 12);
            std::memcpy(q->bitangent, lb, 12);
        }
    }
}

M4 node_matrix(const Json &n) {   // the node's local transform as glTF defines it (column-major), fp32
    if (n.has("matrix")) {
        const Json &a = n.at("matrix");
        if (a.size() != 16) fail("glTF: node.matrix must have 16 numbers");
        M4 r;
        for (size_t i = 0; i < 16; ++i) r.m[i] = (float)a[i].as_num();
        return r;
    }
    float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
    if (n.has("translation")) for (size_t i = 0; i < 3; ++i) t[i] = (float)n.at("translation")[i].as_num();
    if (n.has("rotation")) for (size_t i = 0; i < 4; ++i) q[i] = (float)n.at("rotation")[i].as_num();
    if (n.has("scale")) for (size_t i = 0; i < 3; ++i) s[i] = (float)n.at("scale")[i].as_num();
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    M4 r = identity();
    r.m[0] = (1 - 2 * (y * y + z * z)) * s[0]; r.m[1] = (2 * (x * y + z * w)) * s[0]; r.m[2] = (2 * (x * z - y * w)) * s[0];
    r.m[4] = (2 * (x * y - z * w)) * s[1]; r.m[5] = (1 - 2 * (x * x + z * z)) * s[1]; r.m[6] = (2 * (y * z + x * w)) * s[1];
    r.m[8] = (2 * (x * z + y * w)) * s[2]; r.m[9] = (2 * (y * z - x * w)) * s[2]; r.m[10] = (1 - 2 * (x * x + y * y)) * s[2];
    r.m[12] = t[0]; r.m[13] = t[1]; r.m[14] = t[2];
    return r;
}

std::unique_ptr<ArcticGltf> load(const std::string &path) {
    const std::vector<uint8_t> file = read_file(path);
    std::string text(file.begin(), file.end());
    std::vector<uint8_t> glb_bin;
    if (file.size() >= 4 && !std::memcmp(file.data(), "glTF", 4)) {   // .glb: 12-byte header, then a JSON chunk and an optional BIN chunk
        auto le32 = [&](size_t o) { uint32_t v; if (o + 4 > file.size()) fail("glb: truncated"); std::memcpy(&v, &file[o], 4); return v; };
        if (le32(4) != 2) fail("glb: only container version 2 is supported");
        size_t pos = 12;
        bool have_json = false;
        while (pos + 8 <= file.size()) {
            const uint32_t len = le32(pos), type = le32(pos + 4);
            if (pos + 8 + (size_t)len > file.size()) fail("glb: truncated chunk");
            if (type == 0x4E4F534Au) { text.assign(file.begin() + (long)pos + 8, file.begin() + (long)(pos + 8 + len)); have_json = true; }
            else if (type == 0x004E4942u && glb_bin.empty()) glb_bin.assign(file.begin() + (long)pos + 8, file.begin() + (long)(pos + 8 + len));
            pos += 8 + (size_t)len + ((4 - len % 4) % 4);
        }
        if (!have_json) fail("glb: no JSON chunk");
    }
    const Json doc = JsonParser(text).parse();
    Loader L(doc, dir_of(path));
    L.glb_bin = std::move(glb_bin);
    auto g = std::make_unique<ArcticGltf>();

    // materials (app.cpp:195-294): three images each, fallbacks for the missing ones
    const size_t n_mat = doc.has("materials") ? doc.at("materials").size() : 0;
    for (size_t i = 0; i < n_mat; ++i) {
        const Json &m = doc.at("materials")[i];
        Material out;
        out.img[0] = solid(255, 255, 255); out.img[1] = solid(128, 128, 255); out.img[2] = solid(255, 255, 255);
        if (const Json *pbr = m.find("pbrMetallicRoughness")) {
            if (const Json *t = pbr->find("baseColorTexture")) out.img[0] = L.image_of_texture((size_t)t->at("index").as_int());
            if (const Json *t = pbr->find("metallicRoughnessTexture")) out.img[2] = L.image_of_texture((size_t)t->at("index").as_int());
        }
        if (const Json *t = m.find("normalTexture")) out.img[1] = L.image_of_texture((size_t)t->at("index").as_int());
        g->materials.push_back(std::move(out));
    }
    if (g->materials.empty()) {   // assimp always provides a default material
        Material d;
        d.img[0] = solid(255, 255, 255); d.img[1] = solid(128, 128, 255); d.img[2] = solid(255, 255, 255);
        g->materials.push_back(std::move(d));
    }

    // meshes (app.cpp:296-352): one per primitive, in file order; remember where each glTF mesh starts
    std::vector<size_t> first_of_mesh, count_of_mesh;
    const size_t n_mesh = doc.has("meshes") ? doc.at("meshes").size() : 0;
    for (size_t i = 0; i < n_mesh; ++i) {
        const Json &prims = doc.at("meshes")[i].at("primitives");
        first_of_mesh.push_back(g->meshes.size());
        for (size_t k = 0; k < prims.size(); ++k) {
            const Json &p = prims[k];
            if (p.has("mode") && p.at("mode").as_int() != 4) fail("glTF: only triangle lists (mode 4) are supported");
            const Json &at = p.at("attributes");
            if (!at.has("POSITION")) fail("glTF: primitive without POSITION");
            if (!at.has("NORMAL") || !at.has("TEXCOORD_0")) fail("glTF: primitives need NORMAL and TEXCOORD_0 (load_scene reads both unconditionally)");
            const std::vector<float> pos = L.floats((size_t)at.at("POSITION").as_int(), 3);
            const std::vector<float> nor = L.floats((size_t)at.at("NORMAL").as_int(), 3);
            const std::vector<float> uv = L.floats((size_t)at.at("TEXCOORD_0").as_int(), 2);
            const size_t nv = pos.size() / 3;
            if (nor.size() / 3 != nv || uv.size() / 2 != nv) fail("glTF: attribute counts differ");
            Mesh m;
            m.v.resize(nv);
            for (size_t v = 0; v < nv; ++v) {
                ArcticVertex &o = m.v[v];
                std::memset(&o, 0, sizeof o);
                std::memcpy(o.position, &pos[v * 3], 12);
                std::memcpy(o.normal, &nor[v * 3], 12);
                o.tex_coords[0] = uv[v * 2];
                o.tex_coords[1] = 1.0f - uv[v * 2 + 1];   // aiProcess_FlipUVs
            }
            if (p.has("indices")) m.idx = L.indices((size_t)p.at("indices").as_int());
            else { m.idx.resize(nv); for (size_t v = 0; v < nv; ++v) m.idx[v] = (uint32_t)v; }
            if (m.idx.size() % 3) fail("glTF: index count is not a multiple of 3");
            for (uint32_t ix : m.idx) if (ix >= nv) fail("glTF: index out of range");
            if (at.has("TANGENT")) {   // the importer takes the file's tangents: bitangent = cross(normal, tangent) * w
                const std::vector<float> tan = L.floats((size_t)at.at("TANGENT").as_int(), 4);
                if (tan.size() / 4 != nv) fail("glTF: attribute counts differ");
                for (size_t v = 0; v < nv; ++v) {
                    std::memcpy(m.v[v].tangent, &tan[v * 4], 12);
                    float b[3];
                    cross3(m.v[v].normal, m.v[v].tangent, b);
                    for (int c = 0; c < 3; ++c) m.v[v].bitangent[c] = b[c] * tan[v * 4 + 3];
                }
            } else calc_tangents(m);
            m.material = p.has("material") ? (uint64_t)p.at("material").as_int() : 0;
            if (m.material >= g->materials.size()) fail("glTF: material index out of range");
            g->meshes.push_back(std::move(m));
        }
        count_of_mesh.push_back(prims.size());
    }

    // objects (app.cpp:354-382): depth-first from the root with an explicit stack (children are visited last to first),
    // every node matrix transposed by assimp_to_mat4 (app.cpp:540-564) and accumulated as parent * child
    const size_t n_nodes = doc.has("nodes") ? doc.at("nodes").size() : 0;
    std::vector<size_t> roots;
    if (doc.has("scenes") && doc.at("scenes").size()) {
        const size_t sc = doc.has("scene") ? Loader::index_of(doc.at("scene"), "scene index") : 0;
        const Json &s = doc.at("scenes")[sc];
        if (s.has("nodes")) for (size_t i = 0; i < s.at("nodes").size(); ++i) roots.push_back(Loader::index_of(s.at("nodes")[i], "node index"));
    }
    struct Item { long node; M4 parent; };   // node -1: the synthetic root assimp creates when the scene has several roots
    std::vector<Item> stack;
    std::vector<bool> visited(n_nodes, false);   // the spec makes the node graph a forest: a node reached twice is a cycle or a DAG, both refused
    if (roots.size() == 1) stack.push_back({(long)roots[0], identity()});
    else if (!roots.empty()) stack.push_back({-1, identity()});
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        M4 trs;
        std::vector<size_t> children;
        long mesh = -1;
        if (it.node < 0) { trs = mul(it.parent, identity()); children = roots; }
        else {
            if ((size_t)it.node >= n_nodes) fail("glTF: node index out of range");
            if (visited[(size_t)it.node]) fail("glTF: node graph is not a tree (a node is reachable twice)");
            visited[(size_t)it.node] = true;
            const Json &n = doc.at("nodes")[(size_t)it.node];
            trs = mul(it.parent, transpose(node_matrix(n)));
            if (n.has("children")) for (size_t i = 0; i < n.at("children").size(); ++i) children.push_back(Loader::index_of(n.at("children")[i], "node index"));
            if (n.has("mesh")) mesh = (long)Loader::index_of(n.at("mesh"), "mesh index");
        }
        for (size_t c : children) stack.push_back({(long)c, trs});
        if (mesh >= 0) {
            if ((size_t)mesh >= n_mesh) fail("glTF: mesh index out of range");
            for (size_t k = 0; k < count_of_mesh[(size_t)mesh]; ++k) {
                ArcticObject o;
                std::memset(&o, 0, sizeof o);
                std::memcpy(o.trs, trs.m, sizeof o.trs);
                o.mesh_idx = first_of_mesh[(size_t)mesh] + k;
                g->objects.push_back(o);
            }
        }
    }
    return g;
}

void say(char *err, uint64_t err_len, const char *m) { if (err && err_len) std::snprintf(err, (size_t)err_len, "%s", m); }

}  // namespace

extern "C" {

ArcticGltf *arctic_gltf_load(const char *path, char *err, uint64_t err_len) {
    if (!path) { say(err, err_len, "arctic_gltf_load: null path"); return nullptr; }
    try {
        return load(path).release();
    } catch (const std::exception &e) {
        say(err, err_len, e.what());
        return nullptr;
    }
}
void arctic_gltf_free(ArcticGltf *g) { delete g; }
uint64_t arctic_gltf_material_count(const ArcticGltf *g) { return g ? g->materials.size() : 0; }
uint64_t arctic_gltf_mesh_count(const ArcticGltf *g) { return g ? g->meshes.size() : 0; }
uint64_t arctic_gltf_object_count(const ArcticGltf *g) { return g ? g->objects.size() : 0; }

int arctic_gltf_material_image(const ArcticGltf *g, uint64_t i, int k, const uint8_t **rgba, uint32_t *w, uint32_t *h) {
    if (!g || i >= g->materials.size() || k < 0 || k > 2 || !rgba || !w || !h) return ARCTIC_E_INVALID;
    const Image &im = g->materials[i].img[k];
    *rgba = im.px.data(); *w = im.w; *h = im.h;
    return ARCTIC_OK;
}
int arctic_gltf_mesh(const ArcticGltf *g, uint64_t i, const ArcticVertex **vertices, uint64_t *n_vertices, const uint32_t **indices,
                     uint64_t *n_indices, uint64_t *material) {
    if (!g || i >= g->meshes.size() || !vertices || !n_vertices || !indices || !n_indices || !material) return ARCTIC_E_INVALID;
    const Mesh &m = g->meshes[i];
    *vertices = m.v.data(); *n_vertices = m.v.size(); *indices = m.idx.data(); *n_indices = m.idx.size(); *material = m.material;
    return ARCTIC_OK;
}
const ArcticObject *arctic_gltf_objects(const ArcticGltf *g) { return g && !g->objects.empty() ? g->objects.data() : nullptr; }

int arctic_gltf_upload(const ArcticGltf *g, ArcticRenderer *r) {
    if (!g || !r) return ARCTIC_E_INVALID;
    for (const Material &m : g->materials) {
        int rc = arctic_create_material(r, m.img[0].px.data(), m.img[0].w, m.img[0].h, m.img[1].px.data(), m.img[1].w, m.img[1].h,
                                        m.img[2].px.data(), m.img[2].w, m.img[2].h);
        if (rc < 0) return rc;
    }
    for (const Mesh &m : g->meshes) {
        int rc = arctic_create_mesh(r, m.v.data(), m.v.size(), m.idx.data(), m.idx.size(), m.material);
        if (rc < 0) return rc;
    }
    return ARCTIC_OK;
}

uint8_t *arctic_png_decode(const uint8_t *data, uint64_t size, uint32_t *w, uint32_t *h, char *err, uint64_t err_len) {
    if (!data || !w || !h) { say(err, err_len, "arctic_png_decode: null argument"); return nullptr; }
    try {
        std::vector<uint8_t> px = decode_image(data, (size_t)size, *w, *h);   // PNG or baseline JPEG, by signature
        uint8_t *out = static_cast<uint8_t *>(std::malloc(px.size()));
        if (!out) fail("out of memory");
        std::memcpy(out, px.data(), px.size());
        return out;
    } catch (const std::exception &e) {
        say(err, err_len, e.what());
        return nullptr;
    }
}
void arctic_png_free(uint8_t *p) { std::free(p); }

}  // extern "C"
