// image_io.cpp — dependency-free PNG (stored deflate blocks) and BMP writers; BMP and PNG
// readers (PNG inflate through zlib) for the cubemap faces.
// Counterpart of src/fileio/images.cu:59-68 (extension dispatch),
// src/fileio/pngimage.cu:226-285 and src/fileio/bitmap.cu:95-149.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <zlib.h>

#include "tracer.h"

namespace pt {
namespace {

uint32_t crcTable[256];
void initCrc() {
    static bool done = false;
    if (done) return;
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crcTable[n] = c;
    }
    done = true;
}
uint32_t crc32(const uint8_t *p, size_t n, uint32_t c = 0xFFFFFFFFu) {
    for (size_t i = 0; i < n; i++) c = crcTable[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c;
}
void be32(std::vector<uint8_t> &o, uint32_t v) { o.push_back(v >> 24); o.push_back(v >> 16); o.push_back(v >> 8); o.push_back(v); }
void chunk(std::vector<uint8_t> &out, const char *type, const std::vector<uint8_t> &data) {
    be32(out, (uint32_t)data.size());
    std::vector<uint8_t> td(type, type + 4);
    td.insert(td.end(), data.begin(), data.end());
    out.insert(out.end(), td.begin(), td.end());
    be32(out, crc32(td.data(), td.size()) ^ 0xFFFFFFFFu);
}

bool writePng(const char *path, int w, int h, const unsigned char *rgb) {
    initCrc();
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (1 + (size_t)w * 3));
    for (int y = h - 1; y >= 0; y--) { // flip: buffer row 0 is the bottom row (pngimage.cu:258-268)
        raw.push_back(0);
        raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (uint8_t c : raw) { a = (a + c) % 65521; b = (b + a) % 65521; }
    size_t pos = 0;
    do {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8);
        z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    } while (pos < raw.size());
    be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}

bool writeBmp(const char *path, int w, int h, const unsigned char *rgb) { // bitmap.cu:95-149: BGR, rows padded to 4 B
    const int rowBytes = (w * 3 + 3) & ~3;
    std::vector<uint8_t> out(54 + (size_t)rowBytes * h, 0);
    auto le32 = [&](size_t o, uint32_t v) { out[o] = v; out[o + 1] = v >> 8; out[o + 2] = v >> 16; out[o + 3] = v >> 24; };
    out[0] = 'B'; out[1] = 'M';
    le32(2, (uint32_t)out.size()); le32(10, 54); le32(14, 40); le32(18, (uint32_t)w); le32(22, (uint32_t)h);
    out[26] = 1; out[28] = 24; le32(34, (uint32_t)rowBytes * h);
    for (int y = 0; y < h; y++) // BMP is bottom-up already
        for (int x = 0; x < w; x++) {
            const unsigned char *p = rgb + ((size_t)y * w + x) * 3;
            uint8_t *q = out.data() + 54 + (size_t)y * rowBytes + (size_t)x * 3;
            q[0] = p[2]; q[1] = p[1]; q[2] = p[0];
        }
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}

} // namespace

// ---------------------------------------------------------------- readers
// Counterparts of readBMP (src/fileio/bitmap.cu:17-93) and readPNG
// (src/fileio/pngimage.cu:195-216): RGB8, row 0 = BOTTOM row of the picture.
namespace {

bool readFile(const char *path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    fclose(f);
    return true;
}
uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint32_t rbe32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

// 24-bit uncompressed bottom-up BMP only, like the reference (bitmap.cu:37-41).
bool readBmp(const char *path, int &w, int &h, std::vector<uint8_t> &rgb) {
    std::vector<uint8_t> d;
    if (!readFile(path, d) || d.size() < 54 || d[0] != 'B' || d[1] != 'M') return false;
    const uint32_t off = le32(&d[10]);
    const int32_t bw = (int32_t)le32(&d[18]), bh = (int32_t)le32(&d[22]);
    const unsigned bits = d[28] | (d[29] << 8);
    if (bits != 24 || bw <= 0 || bh <= 0 || bw > 32768 || bh > 32768) return false;
    const size_t pitch = ((size_t)bw * 3 + 3) / 4 * 4;              // rows padded to 4 bytes, bitmap.cu:52-58
    if (off > d.size() || pitch * bh > d.size() - off) return false;
    w = bw; h = bh;
    rgb.resize((size_t)w * h * 3);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {                               // BGR -> RGB, file order kept (bottom-up), bitmap.cu:70-93
            const uint8_t *p = &d[off + pitch * y + 3 * (size_t)x];
            uint8_t *q = &rgb[((size_t)y * w + x) * 3];
            q[0] = p[2]; q[1] = p[1]; q[2] = p[0];
        }
    return true;
}

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// 8-bit, non-interlaced PNG (grey, grey+alpha, RGB, RGBA, palette); alpha is dropped and a
// gAMA chunk is ignored (the reference lets libpng apply it, pngimage.cu:150-155 — files
// written without gAMA, the common case, decode identically).  Rows are flipped so that
// row 0 is the bottom row, pngimage.cu:210-213.
bool readPng(const char *path, int &w, int &h, std::vector<uint8_t> &rgb) {
    std::vector<uint8_t> d;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (!readFile(path, d) || d.size() < 8 + 25 || memcmp(d.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    uint32_t pw = 0, ph = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    while (pos + 12 <= d.size()) {
        const uint32_t len = rbe32(&d[pos]);
        if (len > d.size() - pos - 12) return false;
        const char *type = (const char *)&d[pos + 4];
        const uint8_t *body = &d[pos + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            pw = rbe32(body); ph = rbe32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch || depth != 8 || interlace != 0 || pw == 0 || ph == 0 || pw > 32768 || ph > 32768) return false;
    if (ctype == 3 && plte.size() < 3) return false;
    const size_t pitch = (size_t)pw * ch;
    std::vector<uint8_t> raw((pitch + 1) * ph);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) return false;
    std::vector<uint8_t> img(pitch * ph);
    for (uint32_t y = 0; y < ph; y++) {                             // PNG filters 0-4
        const uint8_t ft = raw[(pitch + 1) * y];
        const uint8_t *in = &raw[(pitch + 1) * y + 1];
        uint8_t *out = &img[pitch * y];
        const uint8_t *up = y ? &img[pitch * (y - 1)] : nullptr;
        for (size_t i = 0; i < pitch; i++) {
            const int a = i >= (size_t)ch ? out[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int v = in[i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: v += paeth(a, b, c); break;
                default: return false;
            }
            out[i] = (uint8_t)v;
        }
    }
    w = (int)pw; h = (int)ph;
    rgb.resize((size_t)w * h * 3);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *p = &img[pitch * (size_t)(h - 1 - y) + (size_t)x * ch];
            uint8_t *q = &rgb[((size_t)y * w + x) * 3];
            if (ctype == 3) {
                const size_t k = (size_t)p[0] * 3;
                if (k + 3 > plte.size()) return false;
                q[0] = plte[k]; q[1] = plte[k + 1]; q[2] = plte[k + 2];
            } else if (ch <= 2) {
                q[0] = q[1] = q[2] = p[0];
            } else {
                q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
            }
        }
    return true;
}

} // namespace

bool readImage(const char *path, int &w, int &h, std::vector<unsigned char> &rgb) { // images.cu:51-57
    std::string p(path);
    size_t dot = p.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : p.substr(dot);
    for (char &c : ext) c = (char)tolower((unsigned char)c);
    if (ext == ".png") return readPng(path, w, h, rgb);
    if (ext == ".bmp") return readBmp(path, w, h, rgb);
    return false;
}

bool writeImage(const char *path, int w, int h, const unsigned char *rgb) { // images.cu:59-68
    std::string p(path);
    size_t dot = p.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : p.substr(dot);
    for (char &c : ext) c = (char)tolower((unsigned char)c);
    return ext == ".png" ? writePng(path, w, h, rgb) : writeBmp(path, w, h, rgb);
}

} // namespace pt
