// Minimal PNG codec for the Image / Write nodes (src/node/image.rs:10-26, src/node/write.rs:5-21,
// read_slot_image src/shared.rs:218-261).  Host-side file I/O only: the decoded interleaved u8
// samples go straight to HBM and are split into f32 planes by from_u8_kernel.
// Supports what `image::open(..).as_flat_samples_u8()` yields for non-interlaced 8-bit gray,
// gray+alpha, RGB, RGBA and 1..8-bit palette files.
#include <zlib.h>

#include <cstdio>
#include <cstring>

#include "kc_runtime.hpp"

namespace kc {

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static int paeth(int a, int b, int c)
{
    const int p = a + b - c;
    const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

int png_read(const std::string &path, std::vector<uint8_t> &px, uint32_t &w, uint32_t &h, int &channels)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) {
        set_error("cannot open " + path);
        return KC_ERR_IO;
    }
    std::vector<uint8_t> data;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) data.insert(data.end(), buf, buf + n);
    std::fclose(f);
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    if (data.size() < 8 || std::memcmp(data.data(), sig, 8) != 0) {
        set_error(path + ": not a PNG");
        return KC_ERR_IMAGE;
    }
    size_t pos = 8;
    std::vector<uint8_t> idat, plte, trns;
    int depth = 0, color = 0, interlace = 0;
    bool have_ihdr = false;
    while (pos + 12 <= data.size()) {
        const uint32_t len = be32(&data[pos]);
        const char *type = (const char *)&data[pos + 4];
        if (pos + 12 + (size_t)len > data.size()) break;
        const uint8_t *body = &data[pos + 8];
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(body);
            h = be32(body + 4);
            depth = body[8];
            color = body[9];
            interlace = body[12];
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    int chans = 0;
    switch (color) {
    case 0: chans = 1; break;
    case 2: chans = 3; break;
    case 3: chans = 1; break;
    case 4: chans = 2; break;
    case 6: chans = 4; break;
    }
    const bool depth_ok = depth == 8 || (color == 3 && (depth == 1 || depth == 2 || depth == 4));
    if (!have_ihdr || chans == 0 || !depth_ok || interlace != 0 || w == 0 || h == 0 || w > 65535 || h > 65535) {
        set_error(path + ": unsupported PNG variant");
        return KC_ERR_IMAGE;
    }
    const size_t stride = ((size_t)w * chans * depth + 7) / 8;
    const size_t bpp = std::max<size_t>(1, (size_t)chans * depth / 8);
    // deflate expands at most 1032:1: a header that promises more than the IDAT bytes can hold is corrupt
    // (and must not drive the allocations below -- a 100-byte file could otherwise ask for 17 GB)
    if ((stride + 1) * h > idat.size() * 1032 + 64) {
        set_error(path + ": corrupt PNG stream (image larger than its data)");
        return KC_ERR_IMAGE;
    }
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), idat.size()) != Z_OK || raw_len != raw.size()) {
        set_error(path + ": corrupt PNG stream");
        return KC_ERR_IMAGE;
    }
    std::vector<uint8_t> lines(stride * h);
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t *in = &raw[(stride + 1) * y + 1];
        uint8_t *cur = &lines[stride * y];
        const uint8_t *prev = y ? &lines[stride * (y - 1)] : zero.data();
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0;
            const int b = prev[i];
            const int c = i >= bpp ? prev[i - bpp] : 0;
            int v = in[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: set_error(path + ": bad PNG filter"); return KC_ERR_IMAGE;
            }
            cur[i] = (uint8_t)v;
        }
    }
    if (color == 3) {
        const bool alpha = !trns.empty();
        channels = alpha ? 4 : 3;
        px.resize((size_t)w * h * channels);
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) {
                uint32_t idx;
                if (depth == 8) {
                    idx = lines[stride * y + x];
                } else {
                    const size_t bit = (size_t)x * depth;
                    idx = (lines[stride * y + bit / 8] >> (8 - depth - bit % 8)) & ((1u << depth) - 1);
                }
                uint8_t *o = &px[((size_t)y * w + x) * channels];
                for (int k = 0; k < 3; ++k) o[k] = (size_t)idx * 3 + k < plte.size() ? plte[(size_t)idx * 3 + k] : 0;
                if (alpha) o[3] = idx < trns.size() ? trns[idx] : 255;
            }
    } else {
        channels = chans;
        px = std::move(lines);
    }
    return KC_OK;
}

int png_write_rgba8(const std::string &path, const uint8_t *px, uint32_t w, uint32_t h)
{
    std::vector<uint8_t> raw(((size_t)w * 4 + 1) * h);
    for (uint32_t y = 0; y < h; ++y) {
        raw[((size_t)w * 4 + 1) * y] = 0;
        std::memcpy(&raw[((size_t)w * 4 + 1) * y + 1], px + (size_t)y * w * 4, (size_t)w * 4);
    }
    uLongf zlen = compressBound(raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), raw.size(), 6) != Z_OK) {
        set_error("PNG deflate failed");
        return KC_ERR_IMAGE;
    }
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) {
        set_error("cannot write " + path);
        return KC_ERR_IO;
    }
    auto put32 = [](uint8_t *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
    auto chunk = [&](const char *type, const uint8_t *body, uint32_t len) {
        uint8_t hdr[8];
        put32(hdr, len);
        std::memcpy(hdr + 4, type, 4);
        std::fwrite(hdr, 1, 8, f);
        if (len) std::fwrite(body, 1, len, f);
        uLong crc = crc32(0L, (const Bytef *)type, 4);
        if (len) crc = crc32(crc, body, len);
        uint8_t c[4];
        put32(c, (uint32_t)crc);
        std::fwrite(c, 1, 4, f);
    };
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    std::fwrite(sig, 1, 8, f);
    uint8_t ihdr[13];
    put32(ihdr, w);
    put32(ihdr + 4, h);
    ihdr[8] = 8;
    ihdr[9] = 6;
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)zlen);
    chunk("IEND", nullptr, 0);
    std::fclose(f);
    return KC_OK;
}

}  // namespace kc
