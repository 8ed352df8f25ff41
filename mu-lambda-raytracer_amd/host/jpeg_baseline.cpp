// host/jpeg_baseline.cpp - a small baseline-JPEG reader for the one data format the path's inputs arrive in.
//
// The reference loads its image texture with `image::open("earthmap.jpg")` (/root/reference/src/worlds.rs:180,441;
// crates image 0.23.14 / jpeg-decoder 0.1.22, not vendored).  This reader covers what that file needs and a bit more:
// baseline sequential DCT (SOF0), 8-bit, 1 or 3 components, Huffman coding, any sampling factors up to 2x2 (chroma is
// replicated, not interpolated), restart intervals.  Progressive / arithmetic / 12-bit files are refused with a
// message.  The inverse DCT is evaluated in double precision and rounded, so results agree with other decoders to
// within their own +-1..2 LSB rounding differences; texture parity on the earth sphere is "unpinned" beyond that
// (SURVEY.md 8c).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_internal.hpp"

namespace {

struct Huff {
    // canonical code lookup: for each length 1..16 the first code, first symbol index and count
    int mincode[17], maxcode[18], valptr[17];
    std::vector<uint8_t> symbols;
    bool defined = false;
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int pred = 0;
    int blocks_w = 0, blocks_h = 0;  // in 8x8 blocks, padded to whole MCUs
    std::vector<uint8_t> plane;      // blocks_w*8 x blocks_h*8
};

struct Decoder {
    const uint8_t* p;
    const uint8_t* end;
    std::string error;
    uint16_t qt[4][64];
    bool qt_defined[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int width = 0, height = 0, restart_interval = 0;
    std::vector<Component> comps;
    // bit reader
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    bool hit_marker = false;

    bool fail(const std::string& m) { if (error.empty()) error = m; return false; }
    int u8() { return p < end ? *p++ : -1; }
    int u16() { const int a = u8(), b = u8(); return (a < 0 || b < 0) ? -1 : ((a << 8) | b); }

    void build(Huff& h, const uint8_t counts[16], const uint8_t* syms, int n) {
        h.symbols.assign(syms, syms + n);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            h.valptr[len] = k;
            h.mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            h.maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        h.maxcode[17] = 0x7fffffff;
        h.defined = true;
    }
    int getbit() {
        if (bitcnt == 0) {
            int b = 0;
            if (!hit_marker) {
                b = u8();
                if (b < 0) { b = 0; hit_marker = true; }
                if (b == 0xFF) {
                    const int b2 = u8();
                    if (b2 != 0) { hit_marker = true; p -= 2; b = 0; }  // a marker: feed zeros, leave it for the caller
                }
            }
            bitbuf = (uint32_t)b;
            bitcnt = 8;
        }
        bitcnt--;
        return (int)((bitbuf >> bitcnt) & 1u);
    }
    int getbits(int n) { int v = 0; for (int i = 0; i < n; i++) v = (v << 1) | getbit(); return v; }
    int decode(const Huff& h) {
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (code << 1) | getbit();
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.symbols[(size_t)(h.valptr[len] + code - h.mincode[len])];
        }
        return -1;
    }
    static int extend(int v, int n) { return n == 0 ? 0 : (v < (1 << (n - 1)) ? v - (1 << n) + 1 : v); }

    bool block(Component& c, int bx, int by) {
        static const int zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                       41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                       30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        double coef[64];
        for (double& x : coef) x = 0.0;
        const int t = decode(dc[c.td]);
        if (t < 0 || t > 11) return fail("bad DC Huffman code");
        c.pred += extend(getbits(t), t);
        coef[0] = (double)c.pred * qt[c.tq][0];
        for (int k = 1; k < 64;) {
            const int rs = decode(ac[c.ta]);
            if (rs < 0) return fail("bad AC Huffman code");
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r == 15) { k += 16; continue; }
                break;  // end of block
            }
            k += r;
            if (k > 63) return fail("AC coefficient index out of range");
            coef[zigzag[k]] = (double)extend(getbits(s), s) * qt[c.tq][k];
            k++;
        }
        // separable inverse DCT in double precision
        static double cosv[8][8];
        static bool ready = false;
        if (!ready) {
            for (int x = 0; x < 8; x++)
                for (int u = 0; u < 8; u++) cosv[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
            ready = true;
        }
        double tmp[64];
        for (int v = 0; v < 8; v++)
            for (int x = 0; x < 8; x++) {
                double s = 0.0;
                for (int u = 0; u < 8; u++) s += cosv[x][u] * coef[v * 8 + u];
                tmp[v * 8 + x] = s;
            }
        const int stride = c.blocks_w * 8;
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) {
                double s = 0.0;
                for (int v = 0; v < 8; v++) s += cosv[y][v] * tmp[v * 8 + x];
                const long q = std::lround(s + 128.0);
                c.plane[(size_t)(by * 8 + y) * stride + (size_t)(bx * 8 + x)] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
            }
        return true;
    }

    bool scan() {
        int hmax = 1, vmax = 1;
        for (const Component& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
        const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
        for (Component& c : comps) {
            c.blocks_w = mcux * c.h; c.blocks_h = mcuy * c.v;
            c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
            c.pred = 0;
            if (!qt_defined[c.tq] || !dc[c.td].defined || !ac[c.ta].defined) return fail("scan uses an undefined table");
        }
        int until_restart = restart_interval;
        for (int my = 0; my < mcuy; my++)
            for (int mx = 0; mx < mcux; mx++) {
                if (restart_interval && until_restart == 0) {
                    // byte-align and consume the RSTn marker
                    bitcnt = 0; hit_marker = false;
                    while (p < end && *p != 0xFF) p++;
                    if (p + 1 < end && p[1] >= 0xD0 && p[1] <= 0xD7) p += 2; else return fail("missing restart marker");
                    for (Component& c : comps) c.pred = 0;
                    until_restart = restart_interval;
                }
                for (Component& c : comps)
                    for (int by = 0; by < c.v; by++)
                        for (int bx = 0; bx < c.h; bx++)
                            if (!block(c, mx * c.h + bx, my * c.v + by)) return false;
                if (restart_interval) until_restart--;
            }
        return true;
    }

    bool run(std::vector<uint8_t>& rgb) {
        if (u16() != 0xFFD8) return fail("not a JPEG file (no SOI)");
        bool got_frame = false;
        for (;;) {
            int m = u8();
            if (m < 0) return fail("unexpected end of file");
            if (m != 0xFF) continue;
            do { m = u8(); } while (m == 0xFF);
            if (m < 0) return fail("unexpected end of file");
            if (m == 0xD9) return fail("end of image before a scan");
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            const int len = u16();
            if (len < 2 || p + (len - 2) > end) return fail("bad segment length");
            const uint8_t* seg = p;
            const uint8_t* seg_end = p + (len - 2);
            p = seg_end;
            // every field below is read only after its bytes are known to lie inside the segment (a truncated or lying
            // segment length must fail, not read past the buffer)
            auto need = [&](ptrdiff_t bytes) { return seg_end - seg >= bytes; };
            if (m == 0xDB) {  // DQT
                while (seg < seg_end) {
                    const int pq = seg[0] >> 4, tq = seg[0] & 15;
                    seg++;
                    if (tq > 3) return fail("bad quantisation table id");
                    if (!need(pq ? 128 : 64)) return fail("bad segment length (DQT)");
                    for (int i = 0; i < 64; i++) {
                        if (pq) { qt[tq][i] = (uint16_t)((seg[0] << 8) | seg[1]); seg += 2; } else { qt[tq][i] = *seg++; }
                    }
                    qt_defined[tq] = true;
                }
            } else if (m == 0xC4) {  // DHT
                while (seg < seg_end) {
                    const int tc = seg[0] >> 4, th = seg[0] & 15;
                    seg++;
                    if (th > 3 || tc > 1) return fail("bad Huffman table id");
                    if (!need(16)) return fail("bad segment length (DHT)");
                    uint8_t counts[16];
                    int n = 0;
                    for (int i = 0; i < 16; i++) { counts[i] = seg[i]; n += seg[i]; }
                    seg += 16;
                    if (seg + n > seg_end || n > 256) return fail("bad Huffman table");
                    build(tc ? ac[th] : dc[th], counts, seg, n);
                    seg += n;
                }
            } else if (m == 0xC0 || m == 0xC1) {  // SOF0 / SOF1 (Huffman, sequential)
                if (!need(6)) return fail("bad segment length (SOF)");
                if (seg[0] != 8) return fail("only 8-bit JPEG is supported");
                height = (seg[1] << 8) | seg[2];
                width = (seg[3] << 8) | seg[4];
                const int n = seg[5];
                if ((n != 1 && n != 3) || width <= 0 || height <= 0) return fail("unsupported component count or size");
                if (!need(6 + 3 * n)) return fail("bad segment length (SOF components)");
                if ((int64_t)width * height > (int64_t)1 << 28) return fail("image too large");
                comps.assign((size_t)n, Component());
                for (int i = 0; i < n; i++) {
                    comps[(size_t)i].id = seg[6 + 3 * i];
                    comps[(size_t)i].h = seg[7 + 3 * i] >> 4;
                    comps[(size_t)i].v = seg[7 + 3 * i] & 15;
                    comps[(size_t)i].tq = seg[8 + 3 * i];
                    if (comps[(size_t)i].h < 1 || comps[(size_t)i].h > 2 || comps[(size_t)i].v < 1 || comps[(size_t)i].v > 2 || comps[(size_t)i].tq > 3)
                        return fail("unsupported sampling factors");
                }
                got_frame = true;
            } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
                return fail("progressive / lossless / arithmetic JPEG is not supported (baseline only)");
            } else if (m == 0xDD) {  // DRI
                if (!need(2)) return fail("bad segment length (DRI)");
                restart_interval = (seg[0] << 8) | seg[1];
            } else if (m == 0xDA) {  // SOS
                if (!got_frame) return fail("scan before frame header");
                if (!need(1)) return fail("bad segment length (SOS)");
                const int n = seg[0];
                if (n != (int)comps.size()) return fail("only single-scan (interleaved) baseline files are supported");
                if (!need(1 + 2 * n + 3)) return fail("bad segment length (SOS components)");
                for (int i = 0; i < n; i++) {
                    bool found = false;
                    for (Component& c : comps)
                        if (c.id == seg[1 + 2 * i]) { c.td = seg[2 + 2 * i] >> 4; c.ta = seg[2 + 2 * i] & 15; found = true; }
                    if (!found || (seg[2 + 2 * i] >> 4) > 3 || (seg[2 + 2 * i] & 15) > 3) return fail("bad scan header");
                }
                bitcnt = 0; hit_marker = false;
                if (!scan()) return false;
                break;
            }
        }
        // to RGB: chroma replicated to full resolution, JFIF YCbCr -> RGB
        int hmax = 1, vmax = 1;
        for (const Component& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
        rgb.resize((size_t)width * height * 3);
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) {
                double v[3] = {0, 128, 128};
                for (size_t i = 0; i < comps.size(); i++) {
                    const Component& c = comps[i];
                    const int sx = x * c.h / hmax, sy = y * c.v / vmax;
                    v[i] = c.plane[(size_t)sy * (size_t)(c.blocks_w * 8) + (size_t)sx];
                }
                double r, g, b;
                if (comps.size() == 1) { r = g = b = v[0]; }
                else {
                    r = v[0] + 1.402 * (v[2] - 128.0);
                    g = v[0] - 0.344136286 * (v[1] - 128.0) - 0.714136286 * (v[2] - 128.0);
                    b = v[0] + 1.772 * (v[1] - 128.0);
                }
                const double out[3] = {r, g, b};
                for (int k = 0; k < 3; k++) {
                    const long q = std::lround(out[k]);
                    rgb[((size_t)y * width + x) * 3 + k] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
                }
            }
        return true;
    }
};

}  // namespace

extern "C" {

int32_t mlrt_decode_jpeg(const uint8_t* data, int64_t size, uint8_t** rgb_out, int32_t* width, int32_t* height) {
    if (!data || size < 4 || !rgb_out || !width || !height) { mlrt::set_last_error("mlrt_decode_jpeg: bad argument"); return MLRT_ERR_INVALID; }
    try {  // (no exception may cross the C ABI: a header announcing 65535 x 65535 pixels must not end in std::bad_alloc)
        Decoder d;
        d.p = data; d.end = data + size;
        std::vector<uint8_t> rgb;
        if (!d.run(rgb)) { mlrt::set_last_error("mlrt_decode_jpeg: " + d.error); return MLRT_ERR_UNSUPPORTED; }
        uint8_t* out = static_cast<uint8_t*>(std::malloc(rgb.size()));
        if (!out) { mlrt::set_last_error("mlrt_decode_jpeg: out of memory"); return MLRT_ERR_INVALID; }
        std::memcpy(out, rgb.data(), rgb.size());
        *rgb_out = out; *width = d.width; *height = d.height;
        return MLRT_OK;
    } catch (const std::exception& e) {
        mlrt::set_last_error(std::string("mlrt_decode_jpeg: ") + e.what());
        return MLRT_ERR_INVALID;
    }
}

int32_t mlrt_decode_jpeg_file(const char* path, uint8_t** rgb_out, int32_t* width, int32_t* height) {
    FILE* f = path ? std::fopen(path, "rb") : nullptr;
    if (!f) { mlrt::set_last_error(std::string("mlrt_decode_jpeg_file: cannot open ") + (path ? path : "(null)")); return MLRT_ERR_INVALID; }
    std::vector<uint8_t> bytes;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    return mlrt_decode_jpeg(bytes.data(), (int64_t)bytes.size(), rgb_out, width, height);
}

void mlrt_free(void* p) { std::free(p); }

}  // extern "C"
