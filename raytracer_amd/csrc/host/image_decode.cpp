// Texture decoding for the host loader: what `image::open(path)?.to_rgb8()` does in the reference
// (src/core/texture.rs:35-44; image 0.24.1 with png 0.17 / jpeg-decoder 0.2.4, Cargo.lock:448,517) for the formats its
// scenes use -- PNG (non-interlaced, 8/16 bit, grey / grey+alpha / RGB / RGBA / palette) and baseline / extended
// sequential Huffman JPEG (8 bit, 1 or 3 components, any sampling factors).  Neither crate is under /root/reference;
// the algorithms are the published ones:
//   * PNG: RFC 2083 filters over RFC 1950/1951 inflate; to_rgb8 replicates grey into RGB, drops alpha, narrows 16-bit
//     samples to their high byte as image's `u16 -> u8` conversion rounds ((v + 128) / 257); gAMA is not applied
//     (png 0.17 applies no gamma unless asked to);
//   * JPEG: ITU T.81 Huffman decoding; the inverse DCT and the YCbCr -> RGB conversion are the fixed-point forms of
//     stb_image (`stbi__idct_block`, 12-bit constants; BT.601 with 20-bit constants), which jpeg-decoder's idct.rs and
//     `ycbcr_to_rgb` follow; chroma of subsampled images is upsampled with libjpeg's triangle filter for 2:1
//     horizontally (h2v1 / h2v2 rows) and replicated otherwise.
// Pinned by tests/test_host_loader.py: texels of scenes/textures/{checker.png,test.jpg} (hand-checked values, and
// the whole of both images against an independent decoder's output within one level), and -- through the renders --
// by the quiet-pixel comparison with the reference's own PNGs (tests/refstats.py).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rth {

namespace {

// ------------------------------------------------------------------ inflate (RFC 1951) inside zlib (RFC 1950)
struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    bool bad = false;
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    uint32_t bits(int k) {
        while (n < k) {
            if (p >= end) {
                bad = true;
                return 0;
            }
            acc |= static_cast<uint32_t>(*p++) << n;
            n += 8;
        }
        const uint32_t v = acc & ((k == 32) ? 0xFFFFFFFFu : ((1u << k) - 1u));
        acc >>= k;
        n -= k;
        return v;
    }
};

struct Huff {
    uint16_t count[16] = {0}, symbol[288] = {0};
    void build(const uint8_t *len, int n) {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; ++i) offs[i + 1] = static_cast<uint16_t>(offs[i] + count[i]);
        for (int i = 0; i < n; ++i)
            if (len[i]) symbol[offs[len[i]]++] = static_cast<uint16_t>(i);
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; ++len) {
            code |= static_cast<int>(br.bits(1));
            if (br.bad) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

// max_out: what the caller can use (a PNG knows its size from IHDR): a stream that inflates beyond it is refused instead of
// being allowed to grow without bound (a few KB of IDAT can describe gigabytes).
bool inflate_zlib(const std::vector<uint8_t> &in, std::vector<uint8_t> *out, size_t max_out, std::string *err) {
    if (in.size() < 6 || (in[0] & 0x0F) != 8 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) {
        *err = "bad zlib header";
        return false;
    }
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    BitReader br(in.data() + 2, in.data() + in.size());
    for (bool last = false; !last;) {
        if (out->size() > max_out) {
            *err = "compressed data inflates beyond the image's size";
            return false;
        }
        last = br.bits(1) != 0;
        const uint32_t type = br.bits(2);
        if (br.bad) break;
        if (type == 0) {
            br.acc = 0;
            br.n = 0;
            if (br.end - br.p < 4) break;
            const uint32_t len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8);
            br.p += 4;
            if ((len ^ 0xFFFFu) != nlen || static_cast<size_t>(br.end - br.p) < len) break;
            out->insert(out->end(), br.p, br.p + len);
            br.p += len;
            if (last) return true;
            continue;
        }
        if (type == 3) break;
        Huff lit, dist;
        uint8_t lens[320];
        if (type == 1) {
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            lit.build(lens, 288);
            for (int i = 0; i < 30; ++i) lens[i] = 5;
            dist.build(lens, 30);
        } else {
            const int nlen = static_cast<int>(br.bits(5)) + 257, ndist = static_cast<int>(br.bits(5)) + 1, ncode = static_cast<int>(br.bits(4)) + 4;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; ++i) cl[order[i]] = static_cast<uint8_t>(br.bits(3));
            if (br.bad || nlen > 286 || ndist > 30) break;
            Huff clh;
            clh.build(cl, 19);
            int idx = 0;
            bool ok = true;
            while (idx < nlen + ndist) {
                const int sym = clh.decode(br);
                if (sym < 0) {
                    ok = false;
                    break;
                }
                if (sym < 16) {
                    lens[idx++] = static_cast<uint8_t>(sym);
                    continue;
                }
                uint8_t prev = 0;
                int rep;
                if (sym == 16) {
                    if (idx == 0) {
                        ok = false;
                        break;
                    }
                    prev = lens[idx - 1];
                    rep = 3 + static_cast<int>(br.bits(2));
                } else if (sym == 17) {
                    rep = 3 + static_cast<int>(br.bits(3));
                } else {
                    rep = 11 + static_cast<int>(br.bits(7));
                }
                if (idx + rep > nlen + ndist) {
                    ok = false;
                    break;
                }
                while (rep--) lens[idx++] = prev;
            }
            if (!ok || br.bad) break;
            lit.build(lens, nlen);
            dist.build(lens + nlen, ndist);
        }
        bool block_ok = false;
        while (true) {
            if (out->size() > max_out) {
                *err = "compressed data inflates beyond the image's size";
                return false;
            }
            const int sym = lit.decode(br);
            if (sym < 0) break;
            if (sym < 256) {
                out->push_back(static_cast<uint8_t>(sym));
                continue;
            }
            if (sym == 256) {
                block_ok = true;
                break;
            }
            if (sym > 285) break;
            const uint32_t len = lbase[sym - 257] + br.bits(lext[sym - 257]);
            const int ds = dist.decode(br);
            if (ds < 0 || ds > 29) break;
            const uint32_t d = dbase[ds] + br.bits(dext[ds]);
            if (br.bad || d > out->size()) break;
            const size_t from = out->size() - d;
            for (uint32_t i = 0; i < len; ++i) out->push_back((*out)[from + i]);
        }
        if (!block_ok) {
            *err = "corrupt deflate stream";
            return false;
        }
        if (last) return true;
    }
    *err = "corrupt or truncated deflate stream";
    return false;
}

uint32_t be32(const uint8_t *p) { return (static_cast<uint32_t>(p[0]) << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

// ------------------------------------------------------------------ PNG
bool decode_png(const std::vector<uint8_t> &f, uint32_t *w_out, uint32_t *h_out, std::vector<uint8_t> *rgb, std::string *err) {
    size_t p = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    while (p + 12 <= f.size()) {
        const uint32_t len = be32(&f[p]);
        const char *tag = reinterpret_cast<const char *>(&f[p + 4]);
        if (p + 12 + len > f.size()) break;
        const uint8_t *d = &f[p + 8];
        if (!std::memcmp(tag, "IHDR", 4) && len >= 13) {
            w = be32(d);
            h = be32(d + 4);
            depth = d[8];
            ctype = d[9];
            interlace = d[12];
        } else if (!std::memcmp(tag, "PLTE", 4)) {
            plte.assign(d, d + len);
        } else if (!std::memcmp(tag, "IDAT", 4)) {
            idat.insert(idat.end(), d, d + len);
        } else if (!std::memcmp(tag, "IEND", 4)) {
            break;
        }
        p += 12 + len;
    }
    if (w == 0 || h == 0 || w > 32768 || h > 32768) {
        *err = "bad PNG header";
        return false;
    }
    if (interlace) {
        *err = "interlaced PNG is not supported";
        return false;
    }
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: *err = "bad PNG colour type"; return false;
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))) || (ctype == 3 && depth == 16)) {
        *err = "unsupported PNG bit depth";
        return false;
    }
    const size_t bpp = static_cast<size_t>(std::max(1, channels * depth / 8));  // filter unit
    const size_t stride = (static_cast<size_t>(w) * channels * depth + 7) / 8;
    std::vector<uint8_t> raw;
    if (!inflate_zlib(idat, &raw, (stride + 1) * h, err)) return false;
    if (raw.size() < (stride + 1) * h) {
        *err = "truncated PNG image data";
        return false;
    }
    std::vector<uint8_t> img(stride * h);
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {  // RFC 2083 section 6: filter types 0-4
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t *src = &raw[(stride + 1) * y + 1];
        uint8_t *cur = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : zero.data();
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } else if (ft != 0) {
                *err = "bad PNG filter type";
                return false;
            }
            cur[i] = static_cast<uint8_t>(src[i] + pred);
        }
    }
    rgb->resize(static_cast<size_t>(w) * h * 3);
    auto sample = [&](const uint8_t *row, size_t index) -> uint32_t {  // index-th sample of a row, as an 8-bit value
        if (depth == 8) return row[index];
        if (depth == 16) return ((static_cast<uint32_t>(row[2 * index]) << 8 | row[2 * index + 1]) + 128u) / 257u;
        const uint32_t per = 8 / depth, v = (row[index / per] >> (8 - depth * (index % per + 1))) & ((1u << depth) - 1u);
        return ctype == 3 ? v : v * 255u / ((1u << depth) - 1u);
    };
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *row = &img[stride * y];
        uint8_t *o = &(*rgb)[static_cast<size_t>(y) * w * 3];
        for (uint32_t x = 0; x < w; ++x, o += 3) {
            if (ctype == 3) {
                const uint32_t i = sample(row, x);
                if (3 * i + 2 >= plte.size()) {
                    *err = "PNG palette index out of range";
                    return false;
                }
                o[0] = plte[3 * i];
                o[1] = plte[3 * i + 1];
                o[2] = plte[3 * i + 2];
            } else if (channels <= 2) {
                o[0] = o[1] = o[2] = static_cast<uint8_t>(sample(row, static_cast<size_t>(x) * channels));
            } else {
                for (int c = 0; c < 3; ++c) o[c] = static_cast<uint8_t>(sample(row, static_cast<size_t>(x) * channels + c));
            }
        }
    }
    *w_out = w;
    *h_out = h;
    return true;
}

// ------------------------------------------------------------------ JPEG (ITU T.81 baseline / extended sequential, Huffman)
struct JHuff {
    uint8_t bits[17] = {0}, vals[256] = {0};
    int mincode[18] = {0}, maxcode[18] = {0}, valptr[18] = {0};
    bool present = false;
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        present = true;
    }
};

struct JBits {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    bool marker = false;  // ran into a marker: feed zeros (T.81 F.2.2.5)
    JBits(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    int bit() {
        if (n == 0) {
            uint32_t byte = 0;
            if (!marker && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    const uint8_t nx = p < end ? *p : 0;
                    if (nx == 0) ++p;  // stuffed zero
                    else {
                        marker = true;
                        --p;
                        byte = 0;
                    }
                }
            }
            acc = byte;
            n = 8;
        }
        --n;
        return static_cast<int>((acc >> n) & 1u);
    }
    int receive(int s) {
        int v = 0;
        for (int i = 0; i < s; ++i) v = (v << 1) | bit();
        return v;
    }
    void reset() {
        acc = 0;
        n = 0;
        marker = false;
    }
};

int jdecode(JBits &br, const JHuff &h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}
int jextend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }

uint8_t clamp8(int x) { return static_cast<uint8_t>(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// stb_image's stbi__idct_block (12-bit fixed point, +128 level shift folded into the row pass), on dequantised coefficients
void idct_block(const int *d, uint8_t *out, int out_stride) {
    auto f2f = [](double x) { return static_cast<int>(x * 4096 + 0.5); };
    int val[64];
#define RT_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                                    \
    int t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                           \
    p2 = s2;                                                                                          \
    p3 = s6;                                                                                          \
    p1 = (p2 + p3) * f2f(0.5411961f);                                                                 \
    t2 = p1 + p3 * f2f(-1.847759065f);                                                                \
    t3 = p1 + p2 * f2f(0.765366865f);                                                                 \
    p2 = s0;                                                                                          \
    p3 = s4;                                                                                          \
    t0 = (p2 + p3) * 4096;                                                                            \
    t1 = (p2 - p3) * 4096;                                                                            \
    x0 = t0 + t3;                                                                                     \
    x3 = t0 - t3;                                                                                     \
    x1 = t1 + t2;                                                                                     \
    x2 = t1 - t2;                                                                                     \
    t0 = s7;                                                                                          \
    t1 = s5;                                                                                          \
    t2 = s3;                                                                                          \
    t3 = s1;                                                                                          \
    p3 = t0 + t2;                                                                                     \
    p4 = t1 + t3;                                                                                     \
    p1 = t0 + t3;                                                                                     \
    p2 = t1 + t2;                                                                                     \
    p5 = (p3 + p4) * f2f(1.175875602f);                                                               \
    t0 = t0 * f2f(0.298631336f);                                                                      \
    t1 = t1 * f2f(2.053119869f);                                                                      \
    t2 = t2 * f2f(3.072711026f);                                                                      \
    t3 = t3 * f2f(1.501321110f);                                                                      \
    p1 = p5 + p1 * f2f(-0.899976223f);                                                                \
    p2 = p5 + p2 * f2f(-2.562915447f);                                                                \
    p3 = p3 * f2f(-1.961570560f);                                                                     \
    p4 = p4 * f2f(-0.390180644f);                                                                     \
    t3 += p1 + p4;                                                                                    \
    t2 += p2 + p3;                                                                                    \
    t1 += p2 + p4;                                                                                    \
    t0 += p1 + p3;
    for (int i = 0; i < 8; ++i) {  // columns
        const int *c = d + i;
        int *v = val + i;
        if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
            const int dc = c[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
        } else {
            RT_IDCT_1D(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56])
            x0 += 512;
            x1 += 512;
            x2 += 512;
            x3 += 512;
            v[0] = (x0 + t3) >> 10;
            v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10;
            v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10;
            v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10;
            v[32] = (x3 - t0) >> 10;
        }
    }
    for (int i = 0; i < 8; ++i) {  // rows
        const int *v = val + 8 * i;
        uint8_t *o = out + i * out_stride;
        RT_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        x0 += 65536 + (128 << 17);
        x1 += 65536 + (128 << 17);
        x2 += 65536 + (128 << 17);
        x3 += 65536 + (128 << 17);
        o[0] = clamp8((x0 + t3) >> 17);
        o[7] = clamp8((x0 - t3) >> 17);
        o[1] = clamp8((x1 + t2) >> 17);
        o[6] = clamp8((x1 - t2) >> 17);
        o[2] = clamp8((x2 + t1) >> 17);
        o[5] = clamp8((x2 - t1) >> 17);
        o[3] = clamp8((x3 + t0) >> 17);
        o[4] = clamp8((x3 - t0) >> 17);
    }
#undef RT_IDCT_1D
}

bool decode_jpeg(const std::vector<uint8_t> &f, uint32_t *w_out, uint32_t *h_out, std::vector<uint8_t> *rgb, std::string *err) {
    static const uint8_t zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                       41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                       30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    uint16_t qt[4][64] = {{0}};
    JHuff hdc[4], hac[4];
    struct Comp {
        int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
        int bw = 0, bh = 0;  // blocks per row / column (padded to whole MCUs)
        std::vector<uint8_t> plane;
        int pred = 0;
    } comp[3];
    int nc = 0, width = 0, height = 0, hmax = 1, vmax = 1, restart = 0, adobe_transform = -1;
    bool have_sof = false;
    size_t p = 2;
    while (p + 4 <= f.size()) {
        if (f[p] != 0xFF) {
            *err = "bad JPEG marker";
            return false;
        }
        const uint8_t m = f[p + 1];
        if (m == 0xFF) {
            ++p;
            continue;
        }
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) {
            p += 2;
            continue;
        }
        const size_t len = (static_cast<size_t>(f[p + 2]) << 8) | f[p + 3];
        if (len < 2 || p + 2 + len > f.size()) break;
        const uint8_t *d = &f[p + 4];
        const size_t n = len - 2;
        if (m == 0xDB) {  // DQT
            for (size_t i = 0; i < n;) {
                const int pq = d[i] >> 4, tq = d[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > n) {
                    *err = "bad DQT";
                    return false;
                }
                for (int k = 0; k < 64; ++k) {
                    qt[tq][zigzag[k]] = pq ? static_cast<uint16_t>((d[i] << 8) | d[i + 1]) : d[i];
                    i += pq ? 2 : 1;
                }
            }
        } else if (m == 0xC4) {  // DHT
            for (size_t i = 0; i + 17 <= n;) {
                const int tc = d[i] >> 4, th = d[i] & 15;
                if (tc > 1 || th > 3) {
                    *err = "bad DHT";
                    return false;
                }
                JHuff &h = tc ? hac[th] : hdc[th];
                int total = 0;
                for (int l = 1; l <= 16; ++l) {
                    h.bits[l] = d[i + l];
                    total += d[i + l];
                }
                i += 17;
                if (total > 256 || i + total > n) {
                    *err = "bad DHT";
                    return false;
                }
                std::memcpy(h.vals, d + i, static_cast<size_t>(total));
                i += static_cast<size_t>(total);
                h.build();
            }
        } else if (m == 0xC0 || m == 0xC1) {  // SOF0 / SOF1
            if (n < 6 || d[0] != 8) {
                *err = "only 8-bit JPEG is supported";
                return false;
            }
            height = (d[1] << 8) | d[2];
            width = (d[3] << 8) | d[4];
            nc = d[5];
            if ((nc != 1 && nc != 3) || n < 6 + 3 * static_cast<size_t>(nc) || width <= 0 || height <= 0) {
                *err = "unsupported JPEG frame (components)";
                return false;
            }
            // the planes are allocated from the header before any entropy data is seen: a frame needs at least a bit of
            // data per 8x8 block (a 65535 x 65535 header on a few bytes would ask for 12 GB)
            if (static_cast<uint64_t>(width) * height / 64 / 8 > f.size()) {
                *err = "JPEG frame size does not fit the file's size";
                return false;
            }
            for (int c = 0; c < nc; ++c) {
                comp[c].id = d[6 + 3 * c];
                comp[c].h = d[7 + 3 * c] >> 4;
                comp[c].v = d[7 + 3 * c] & 15;
                comp[c].tq = d[8 + 3 * c] & 3;
                if (comp[c].h < 1 || comp[c].h > 4 || comp[c].v < 1 || comp[c].v > 4) {
                    *err = "bad JPEG sampling factors";
                    return false;
                }
                hmax = std::max(hmax, comp[c].h);
                vmax = std::max(vmax, comp[c].v);
            }
            have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            *err = "progressive / lossless / arithmetic JPEG is not supported (baseline and extended sequential Huffman only)";
            return false;
        } else if (m == 0xDD && n >= 2) {
            restart = (d[0] << 8) | d[1];
        } else if (m == 0xEE && n >= 12 && !std::memcmp(d, "Adobe", 5)) {
            adobe_transform = d[11];
        } else if (m == 0xDA) {  // SOS: the one scan of a sequential image
            if (!have_sof) {
                *err = "JPEG scan before frame header";
                return false;
            }
            const int ns = d[0];
            if (ns != nc || n < 1 + 2 * static_cast<size_t>(ns) + 3) {
                *err = "non-interleaved multi-scan JPEG is not supported";
                return false;
            }
            for (int s = 0; s < ns; ++s) {
                int c = 0;
                while (c < nc && comp[c].id != d[1 + 2 * s]) ++c;
                if (c == nc) {
                    *err = "bad JPEG scan header";
                    return false;
                }
                comp[c].td = d[2 + 2 * s] >> 4;
                comp[c].ta = d[2 + 2 * s] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3 || !hdc[comp[c].td].present || !hac[comp[c].ta].present) {
                    *err = "JPEG scan refers to a missing Huffman table";
                    return false;
                }
            }
            const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
            if (nc == 1) {  // a single-component scan is not interleaved: one block per MCU
                comp[0].h = comp[0].v = hmax = vmax = 1;
            }
            const int mx = nc == 1 ? (width + 7) / 8 : mcux, my = nc == 1 ? (height + 7) / 8 : mcuy;
            for (int c = 0; c < nc; ++c) {
                comp[c].bw = mx * comp[c].h;
                comp[c].bh = my * comp[c].v;
                comp[c].plane.assign(static_cast<size_t>(comp[c].bw) * 8 * comp[c].bh * 8, 0);
            }
            JBits br(&f[p + 2 + len], f.data() + f.size());
            int todo = restart;
            for (int y = 0; y < my; ++y)
                for (int x = 0; x < mx; ++x) {
                    for (int c = 0; c < nc; ++c)
                        for (int by = 0; by < comp[c].v; ++by)
                            for (int bx = 0; bx < comp[c].h; ++bx) {
                                int blk[64] = {0};
                                const int t = jdecode(br, hdc[comp[c].td]);
                                if (t < 0 || t > 11) {
                                    *err = "corrupt JPEG entropy data";
                                    return false;
                                }
                                comp[c].pred += jextend(br.receive(t), t);
                                if (comp[c].pred < -32768 || comp[c].pred > 32767) {  // a DC value is 11 bits in a baseline file (T.81 F.1.2.1)
                                    *err = "corrupt JPEG entropy data (DC out of range)";
                                    return false;
                                }
                                const uint16_t *q = qt[comp[c].tq];
                                // a dequantised coefficient of 8-bit samples is within +-1,200; beyond +-16,384 (corrupt data, 16-bit
                                // tables) the fixed-point IDCT's int products would overflow
                                auto dequant = [](int v, int qv) { return std::max(-16384, std::min(16384, v * qv)); };
                                blk[0] = dequant(comp[c].pred, q[0]);
                                for (int k = 1; k < 64;) {
                                    const int rs = jdecode(br, hac[comp[c].ta]);
                                    if (rs < 0) {
                                        *err = "corrupt JPEG entropy data";
                                        return false;
                                    }
                                    const int r = rs >> 4, s = rs & 15;
                                    if (s == 0) {
                                        if (r != 15) break;  // end of block
                                        k += 16;
                                        continue;
                                    }
                                    k += r;
                                    if (k > 63) {
                                        *err = "corrupt JPEG entropy data";
                                        return false;
                                    }
                                    blk[zigzag[k]] = dequant(jextend(br.receive(s), s), q[zigzag[k]]);
                                    ++k;
                                }
                                const int px = (x * comp[c].h + bx) * 8, py = (y * comp[c].v + by) * 8, stride = comp[c].bw * 8;
                                idct_block(blk, &comp[c].plane[static_cast<size_t>(py) * stride + px], stride);
                            }
                    // the reader ran into a marker (it feeds zeros from there on) with MCUs of this interval still to come: the
                    // file is truncated -- stop, instead of decoding up to 67 million MCUs of zeros for a 65535 x 65535 header
                    if (br.marker && !(y == my - 1 && x == mx - 1) && !(restart && todo == 1)) {
                        *err = "truncated JPEG entropy data";
                        return false;
                    }
                    if (restart && --todo == 0 && !(y == my - 1 && x == mx - 1)) {  // RSTn: byte-align, skip the marker, reset predictors
                        const uint8_t *q = br.p;
                        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                        if (q + 1 >= br.end) {
                            *err = "missing JPEG restart marker";
                            return false;
                        }
                        br.p = q + 2;
                        br.reset();
                        for (int c = 0; c < nc; ++c) comp[c].pred = 0;
                        todo = restart;
                    }
                }
            // ---- to RGB
            rgb->resize(static_cast<size_t>(width) * height * 3);
            std::vector<uint8_t> rows[3];
            for (int y = 0; y < height; ++y) {
                const uint8_t *src[3] = {nullptr, nullptr, nullptr};
                for (int c = 0; c < nc; ++c) {
                    const Comp &k = comp[c];
                    const int stride = k.bw * 8, cw = (width * k.h + hmax - 1) / hmax;
                    const uint8_t *row = &k.plane[static_cast<size_t>(y * k.v / vmax) * stride];
                    if (k.h == hmax) {
                        src[c] = row;
                        continue;
                    }
                    rows[c].resize(static_cast<size_t>(width) + 2);
                    if (hmax == 2 * k.h) {  // 2:1 horizontally: libjpeg's h2v1 "fancy" triangle filter (3/4, 1/4)
                        for (int i = 0; i < cw; ++i) {
                            const int cur = row[i], prev = row[i > 0 ? i - 1 : 0], next = row[i + 1 < cw ? i + 1 : cw - 1];
                            if (2 * i < width) rows[c][2 * i] = static_cast<uint8_t>((3 * cur + prev + 2) >> 2);
                            if (2 * i + 1 < width) rows[c][2 * i + 1] = static_cast<uint8_t>((3 * cur + next + 2) >> 2);
                        }
                    } else {
                        for (int i = 0; i < width; ++i) rows[c][i] = row[std::min(cw - 1, i * k.h / hmax)];
                    }
                    src[c] = rows[c].data();
                }
                uint8_t *o = &(*rgb)[static_cast<size_t>(y) * width * 3];
                for (int x = 0; x < width; ++x, o += 3) {
                    if (nc == 1) {
                        o[0] = o[1] = o[2] = src[0][x];
                    } else if (adobe_transform == 0) {  // Adobe marker says the components are RGB already
                        o[0] = src[0][x];
                        o[1] = src[1][x];
                        o[2] = src[2][x];
                    } else {  // ITU-R BT.601, 20-bit fixed point, round to nearest
                        auto f2f = [](double v) { return static_cast<int>(v * 1048576.0 + 0.5); };
                        const int yy = (static_cast<int>(src[0][x]) << 20) + (1 << 19), cb = src[1][x] - 128, cr = src[2][x] - 128;
                        o[0] = clamp8((yy + f2f(1.40200) * cr) >> 20);
                        o[1] = clamp8((yy - f2f(0.34414) * cb - f2f(0.71414) * cr) >> 20);
                        o[2] = clamp8((yy + f2f(1.77200) * cb) >> 20);
                    }
                }
            }
            *w_out = static_cast<uint32_t>(width);
            *h_out = static_cast<uint32_t>(height);
            return true;
        }
        p += 2 + len;
    }
    *err = "no JPEG scan found";
    return false;
}

}  // namespace

// image::open(path).to_rgb8(): format by content.
bool decode_image(const std::vector<uint8_t> &file, uint32_t *w, uint32_t *h, std::vector<uint8_t> *rgb, std::string *err) {
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() >= 8 && !std::memcmp(file.data(), png_sig, 8)) return decode_png(file, w, h, rgb, err);
    if (file.size() >= 3 && file[0] == 0xFF && file[1] == 0xD8 && file[2] == 0xFF) return decode_jpeg(file, w, h, rgb, err);
    *err = "unsupported image format (PNG and baseline JPEG are decoded)";
    return false;
}

}  // namespace rth
