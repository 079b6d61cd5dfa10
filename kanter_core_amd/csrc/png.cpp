// Minimal PNG codec for the Image / Write nodes (src/node/image.rs:10-26, src/node/write.rs:5-21,
// read_slot_image src/shared.rs:218-261).  Host-side file I/O only: the decoded interleaved u8
// samples go straight to HBM and are split into f32 planes by from_u8_kernel.
// Supports what `image::open(..).as_flat_samples_u8()` yields for 8-bit gray, gray+alpha, RGB, RGBA, 1/2/4-bit gray
// (scaled to 8 bits), 1..8-bit palette files, a tRNS colour key on gray / RGB files (-> alpha channel), plain and
// Adam7-interlaced.  16-bit files are refused (the reference panics on them, src/shared.rs:17).
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
    const bool depth_ok = depth == 8 || ((color == 3 || color == 0) && (depth == 1 || depth == 2 || depth == 4));
    if (!have_ihdr || chans == 0 || !depth_ok || interlace > 1 || w == 0 || h == 0 || w > 65535 || h > 65535) {
        // 16-bit files: the reference opens them and then panics in deconstruct_image (as_flat_samples_u8().unwrap(),
        // src/shared.rs:17); here they are an error like any other file that cannot be read
        set_error(path + ": unsupported PNG variant");
        return KC_ERR_IMAGE;
    }
    // Adam7: seven reduced images, each filtered on its own (x0, y0, dx, dy per pass); non-interlaced = one pass
    static const uint32_t adam7[7][4] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
    static const uint32_t whole[1][4] = { { 0, 0, 1, 1 } };
    const uint32_t(*passes)[4] = interlace ? adam7 : whole;
    const int n_pass = interlace ? 7 : 1;
    const size_t bpp = std::max<size_t>(1, (size_t)chans * depth / 8);
    size_t raw_bytes = 0;
    for (int pi = 0; pi < n_pass; ++pi) {
        const uint32_t pw = (w - passes[pi][0] + passes[pi][2] - 1) / passes[pi][2], ph = (h - passes[pi][1] + passes[pi][3] - 1) / passes[pi][3];
        if (w <= passes[pi][0] || h <= passes[pi][1] || pw == 0 || ph == 0) continue;
        raw_bytes += (((size_t)pw * chans * depth + 7) / 8 + 1) * ph;
    }
    // deflate expands at most 1032:1: a header that promises more than the IDAT bytes can hold is corrupt
    // (and must not drive the allocations below -- a 100-byte file could otherwise ask for 17 GB)
    if (raw_bytes > idat.size() * 1032 + 64) {
        set_error(path + ": corrupt PNG stream (image larger than its data)");
        return KC_ERR_IMAGE;
    }
    std::vector<uint8_t> raw(raw_bytes);
    uLongf raw_len = raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), idat.size()) != Z_OK || raw_len != raw.size()) {
        set_error(path + ": corrupt PNG stream");
        return KC_ERR_IMAGE;
    }
    // samples[y][x][c]: the file's samples at their own depth (gray / index 0 .. 2^depth - 1, or 8-bit)
    std::vector<uint8_t> samples((size_t)w * h * chans);
    size_t rp = 0;
    for (int pi = 0; pi < n_pass; ++pi) {
        const uint32_t px0 = passes[pi][0], py0 = passes[pi][1], dx = passes[pi][2], dy = passes[pi][3];
        if (w <= px0 || h <= py0) continue;
        const uint32_t pw = (w - px0 + dx - 1) / dx, ph = (h - py0 + dy - 1) / dy;
        const size_t stride = ((size_t)pw * chans * depth + 7) / 8;
        std::vector<uint8_t> cur(stride), prev(stride, 0);
        for (uint32_t y = 0; y < ph; ++y) {
            const uint8_t ft = raw[rp];
            const uint8_t *in = &raw[rp + 1];
            rp += stride + 1;
            for (size_t i = 0; i < stride; ++i) {
                const int fa = i >= bpp ? cur[i - bpp] : 0;
                const int fb = prev[i];
                const int fc = i >= bpp ? prev[i - bpp] : 0;
                int v = in[i];
                switch (ft) {
                case 0: break;
                case 1: v += fa; break;
                case 2: v += fb; break;
                case 3: v += (fa + fb) >> 1; break;
                case 4: v += paeth(fa, fb, fc); break;
                default: set_error(path + ": bad PNG filter"); return KC_ERR_IMAGE;
                }
                cur[i] = (uint8_t)v;
            }
            uint8_t *orow = &samples[((size_t)(py0 + y * dy) * w) * chans];
            for (uint32_t x = 0; x < pw; ++x) {
                uint8_t *o = orow + (size_t)(px0 + x * dx) * chans;
                if (depth == 8) {
                    for (int k = 0; k < chans; ++k) o[k] = cur[(size_t)x * chans + k];
                } else {
                    const size_t bit = (size_t)x * depth;
                    o[0] = (cur[bit / 8] >> (8 - depth - bit % 8)) & ((1u << depth) - 1);
                }
            }
            prev.swap(cur);
        }
    }
    // to what image::open(..).as_flat_samples_u8() holds: palette -> RGB(A), low-bit gray scaled to 8 bits, a tRNS
    // colour key on gray / RGB files -> an alpha channel (0 where the pixel equals the key, 255 elsewhere)
    const size_t npx = (size_t)w * h;
    if (color == 3) {
        const bool alpha = !trns.empty();
        channels = alpha ? 4 : 3;
        px.resize(npx * channels);
        for (size_t i = 0; i < npx; ++i) {
            const uint32_t idx = samples[i];
            uint8_t *o = &px[i * channels];
            for (int k = 0; k < 3; ++k) o[k] = (size_t)idx * 3 + k < plte.size() ? plte[(size_t)idx * 3 + k] : 0;
            if (alpha) o[3] = idx < trns.size() ? trns[idx] : 255;
        }
    } else if (color == 0) {
        const bool alpha = trns.size() >= 2;
        const uint32_t key = alpha ? (((uint32_t)trns[0] << 8) | trns[1]) : 0;
        const uint32_t scale = 255u / ((1u << depth) - 1u);  // 255, 85, 17, 1
        channels = alpha ? 2 : 1;
        px.resize(npx * channels);
        for (size_t i = 0; i < npx; ++i) {
            px[i * channels] = (uint8_t)(samples[i] * scale);
            if (alpha) px[i * channels + 1] = samples[i] == key ? 0 : 255;
        }
    } else if (color == 2 && trns.size() >= 6) {
        const uint32_t kr = ((uint32_t)trns[0] << 8) | trns[1], kg = ((uint32_t)trns[2] << 8) | trns[3], kb = ((uint32_t)trns[4] << 8) | trns[5];
        channels = 4;
        px.resize(npx * 4);
        for (size_t i = 0; i < npx; ++i) {
            const uint8_t *p3 = &samples[i * 3];
            uint8_t *o = &px[i * 4];
            o[0] = p3[0];
            o[1] = p3[1];
            o[2] = p3[2];
            o[3] = (p3[0] == kr && p3[1] == kg && p3[2] == kb) ? 0 : 255;
        }
    } else {
        channels = chans;
        px = std::move(samples);
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
