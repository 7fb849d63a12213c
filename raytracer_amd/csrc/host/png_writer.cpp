// RGBA8 -> PNG (what `image.save(filename)` does at src/main.rs:73).
// Uncompressed ("stored") deflate blocks: every PNG reader accepts them and the
// writer needs no external library.
#include "../../../include/mi355rt_host.h"

#include <cstdio>
#include <vector>

namespace {

uint32_t crc_table[256];
bool crc_ready = false;

void crc_init() {
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    crc_ready = true;
}

uint32_t crc_update(uint32_t crc, const uint8_t *buf, size_t len) {
    for (size_t i = 0; i < len; ++i) crc = crc_table[(crc ^ buf[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
}

void put_u32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back(static_cast<uint8_t>(x >> 24));
    v.push_back(static_cast<uint8_t>(x >> 16));
    v.push_back(static_cast<uint8_t>(x >> 8));
    v.push_back(static_cast<uint8_t>(x));
}

void put_chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &data) {
    put_u32(out, static_cast<uint32_t>(data.size()));
    size_t start = out.size();
    for (int i = 0; i < 4; ++i) out.push_back(static_cast<uint8_t>(type[i]));
    out.insert(out.end(), data.begin(), data.end());
    uint32_t crc = crc_update(0xFFFFFFFFu, out.data() + start, out.size() - start) ^ 0xFFFFFFFFu;
    put_u32(out, crc);
}

}  // namespace

extern "C" int rth_write_png(const char *path, uint32_t w, uint32_t h, const uint8_t *rgba8) {
    if (!path || !rgba8 || w == 0 || h == 0) return RT_ERR_INVALID;
    if (!crc_ready) crc_init();

    std::vector<uint8_t> raw;
    raw.reserve(static_cast<size_t>(h) * (static_cast<size_t>(w) * 4 + 1));
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0);  // filter: none
        const uint8_t *row = rgba8 + static_cast<size_t>(y) * w * 4;
        raw.insert(raw.end(), row, row + static_cast<size_t>(w) * 4);
    }

    std::vector<uint8_t> z;
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;  // adler32
    size_t pos = 0;
    while (pos < raw.size()) {
        size_t n = raw.size() - pos;
        if (n > 65535) n = 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back(static_cast<uint8_t>(n & 0xFF));
        z.push_back(static_cast<uint8_t>(n >> 8));
        z.push_back(static_cast<uint8_t>(~n & 0xFF));
        z.push_back(static_cast<uint8_t>((~n >> 8) & 0xFF));
        for (size_t i = 0; i < n; ++i) {
            a = (a + raw[pos + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        z.insert(z.end(), raw.begin() + static_cast<long>(pos), raw.begin() + static_cast<long>(pos + n));
        pos += n;
    }
    put_u32(z, (b << 16) | a);

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_u32(ihdr, w);
    put_u32(ihdr, h);
    ihdr.push_back(8);  // bit depth
    ihdr.push_back(6);  // RGBA
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", z);
    put_chunk(out, "IEND", {});

    FILE *f = std::fopen(path, "wb");
    if (!f) return RT_ERR_IO;
    size_t written = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    return written == out.size() ? RT_OK : RT_ERR_IO;
}
