// image_io.cpp — dependency-free PNG (stored deflate blocks) and BMP writers.
// Counterpart of src/fileio/images.cu:59-68 (extension dispatch),
// src/fileio/pngimage.cu:226-285 and src/fileio/bitmap.cu:95-149.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

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

bool writeImage(const char *path, int w, int h, const unsigned char *rgb) { // images.cu:59-68
    std::string p(path);
    size_t dot = p.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : p.substr(dot);
    for (char &c : ext) c = (char)tolower((unsigned char)c);
    return ext == ".png" ? writePng(path, w, h, rgb) : writeBmp(path, w, h, rgb);
}

} // namespace pt
