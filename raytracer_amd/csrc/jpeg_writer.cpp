// Baseline JPEG/JFIF writer for the frame driver's output (host code, no device work).
//
// The reference writes its result with a third-party encoder (src/toojpeg.cpp) called from save_jpg
// (src/main.cpp:148-180): RGB, quality 90, no chroma subsampling (YCbCr 4:4:4), comment "Render Result".  This file is an
// independent implementation of the same container and coding choices from the JPEG standard (ITU-T T.81): the Annex K
// quantisation tables scaled by the usual quality rule, the Annex K Huffman tables, an 8x8 forward DCT in float, one
// scan with interleaved Y, Cb, Cr blocks.  Decoders reproduce the same picture within the quantisation error; the
// byte stream is not the third-party encoder's.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"

namespace {

const uint8_t kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
// T.81 Annex K.1, in natural (row-major) order
const uint8_t kLumaQ[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                            69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55,  64,
                            81, 104, 113, 92, 49, 64,  78,  87,  103, 121, 120, 101, 72, 92, 95,  98,  112, 100, 103, 99};
const uint8_t kChromaQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                              99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
// T.81 Annex K.3: number of codes of each length 1..16, then the symbols in code order
const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct Huff { uint16_t code[256]; uint8_t len[256]; };

void build_huff(const uint8_t bits[16], const uint8_t* vals, Huff* h) {
    memset(h, 0, sizeof(*h));
    uint16_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l - 1]; ++i, ++k) { h->code[vals[k]] = code++; h->len[vals[k]] = (uint8_t)l; }
        code <<= 1;
    }
}

struct BitWriter {
    std::vector<uint8_t>* out; uint32_t acc = 0; int n = 0;
    void put(uint32_t v, int bits) {
        acc = (acc << bits) | (v & ((1u << bits) - 1u)); n += bits;
        while (n >= 8) {
            const uint8_t b = (uint8_t)(acc >> (n - 8));
            out->push_back(b);
            if (b == 0xFF) out->push_back(0);     // byte stuffing
            n -= 8;
        }
    }
    void flush() { if (n > 0) put(0x7F, 8 - n); }  // pad with ones
};

void marker(std::vector<uint8_t>& o, uint8_t m, size_t payload) {
    o.push_back(0xFF); o.push_back(m);
    o.push_back((uint8_t)((payload + 2) >> 8)); o.push_back((uint8_t)((payload + 2) & 0xFF));
}

// 8x8 forward DCT (separable, float), input level-shifted samples, output in natural order
struct DctBasis {
    float c[8][8];
    DctBasis() {
        for (int u = 0; u < 8; ++u)
            for (int x = 0; x < 8; ++x) c[u][x] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.f);
    }
};
void fdct8x8(const float in[64], float out[64]) {
    static const DctBasis basis;          // a function-local static is initialised once, also with several threads encoding
    const float (*c)[8] = basis.c;
    float tmp[64];
    for (int y = 0; y < 8; ++y)
        for (int u = 0; u < 8; ++u) { float s = 0; for (int x = 0; x < 8; ++x) s += in[y * 8 + x] * c[u][x]; tmp[y * 8 + u] = s; }
    for (int v = 0; v < 8; ++v)
        for (int u = 0; u < 8; ++u) { float s = 0; for (int y = 0; y < 8; ++y) s += tmp[y * 8 + u] * c[v][y]; out[v * 8 + u] = s; }
}

int bit_size(int v) { v = v < 0 ? -v : v; int n = 0; while (v) { ++n; v >>= 1; } return n; }

void encode_block(BitWriter& bw, const float px[64], const uint8_t q[64], const Huff& dc, const Huff& ac, int* last_dc) {
    float f[64];
    fdct8x8(px, f);
    int z[64];
    for (int i = 0; i < 64; ++i) z[i] = (int)std::lrintf(f[kZigZag[i]] / q[kZigZag[i]]);
    const int diff = z[0] - *last_dc; *last_dc = z[0];
    int s = bit_size(diff);
    bw.put(dc.code[s], dc.len[s]);
    if (s) bw.put((uint32_t)(diff < 0 ? diff - 1 : diff), s);
    int last = 63;
    while (last > 0 && z[last] == 0) --last;
    int run = 0;
    for (int i = 1; i <= last; ++i) {
        if (z[i] == 0) { ++run; continue; }
        while (run >= 16) { bw.put(ac.code[0xF0], ac.len[0xF0]); run -= 16; }
        s = bit_size(z[i]);
        const int sym = (run << 4) | s;
        bw.put(ac.code[sym], ac.len[sym]);
        bw.put((uint32_t)(z[i] < 0 ? z[i] - 1 : z[i]), s);
        run = 0;
    }
    if (last < 63) bw.put(ac.code[0], ac.len[0]);   // end of block
}

int encode(const uint8_t* rgb, int width, int height, int quality, const char* comment, std::vector<uint8_t>& o) {
    if (!rgb || width <= 0 || height <= 0 || width > 65535 || height > 65535) return RT_ERR_INVALID;
    try {
        quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
        const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
        uint8_t ql[64], qc[64];
        for (int i = 0; i < 64; ++i) {
            int a = (kLumaQ[i] * scale + 50) / 100, b = (kChromaQ[i] * scale + 50) / 100;
            ql[i] = (uint8_t)(a < 1 ? 1 : (a > 255 ? 255 : a)); qc[i] = (uint8_t)(b < 1 ? 1 : (b > 255 ? 255 : b));
        }
        o.clear();
        o.reserve((size_t)width * height / 2 + 1024);
        o.push_back(0xFF); o.push_back(0xD8);                                    // SOI
        marker(o, 0xE0, 14);                                                     // APP0 / JFIF 1.1, no density, no thumbnail
        const uint8_t jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
        o.insert(o.end(), jfif, jfif + 14);
        if (comment && *comment) {
            const size_t n = strlen(comment);
            if (n > 65000) return RT_ERR_INVALID;
            marker(o, 0xFE, n);
            o.insert(o.end(), comment, comment + n);
        }
        marker(o, 0xDB, 2 * 65);                                                 // DQT: two 8-bit tables, zig-zag order
        o.push_back(0); for (int i = 0; i < 64; ++i) o.push_back(ql[kZigZag[i]]);
        o.push_back(1); for (int i = 0; i < 64; ++i) o.push_back(qc[kZigZag[i]]);
        marker(o, 0xC0, 15);                                                     // SOF0: 8 bit, 3 components, all 1x1 (4:4:4)
        o.push_back(8); o.push_back((uint8_t)(height >> 8)); o.push_back((uint8_t)height); o.push_back((uint8_t)(width >> 8)); o.push_back((uint8_t)width);
        o.push_back(3);
        for (int c = 0; c < 3; ++c) { o.push_back((uint8_t)(c + 1)); o.push_back(0x11); o.push_back(c ? 1 : 0); }
        marker(o, 0xC4, (1 + 16 + 12) * 2 + (1 + 16 + 162) * 2);                 // DHT: DC/AC x luma/chroma
        const struct { uint8_t id; const uint8_t* bits; const uint8_t* vals; int n; } tabs[4] = {
            {0x00, kDcLumaBits, kDcVals, 12}, {0x10, kAcLumaBits, kAcLumaVals, 162}, {0x01, kDcChromaBits, kDcVals, 12}, {0x11, kAcChromaBits, kAcChromaVals, 162}};
        for (const auto& t : tabs) { o.push_back(t.id); o.insert(o.end(), t.bits, t.bits + 16); o.insert(o.end(), t.vals, t.vals + t.n); }
        marker(o, 0xDA, 10);                                                     // SOS
        o.push_back(3);
        for (int c = 0; c < 3; ++c) { o.push_back((uint8_t)(c + 1)); o.push_back(c ? 0x11 : 0x00); }
        o.push_back(0); o.push_back(63); o.push_back(0);

        Huff dcl, acl, dcc, acc_;
        build_huff(kDcLumaBits, kDcVals, &dcl); build_huff(kAcLumaBits, kAcLumaVals, &acl);
        build_huff(kDcChromaBits, kDcVals, &dcc); build_huff(kAcChromaBits, kAcChromaVals, &acc_);
        BitWriter bw; bw.out = &o;
        int last[3] = {0, 0, 0};
        float Y[64], Cb[64], Cr[64];
        for (int by = 0; by < height; by += 8)
            for (int bx = 0; bx < width; bx += 8) {
                for (int y = 0; y < 8; ++y)
                    for (int x = 0; x < 8; ++x) {
                        const int sy = by + y < height ? by + y : height - 1, sx = bx + x < width ? bx + x : width - 1;   // edge replication
                        const uint8_t* p = rgb + ((size_t)sy * width + sx) * 3;
                        const float r = p[0], g = p[1], b = p[2];
                        Y[y * 8 + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.f;
                        Cb[y * 8 + x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
                        Cr[y * 8 + x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
                    }
                encode_block(bw, Y, ql, dcl, acl, &last[0]);
                encode_block(bw, Cb, qc, dcc, acc_, &last[1]);
                encode_block(bw, Cr, qc, dcc, acc_, &last[2]);
            }
        bw.flush();
        o.push_back(0xFF); o.push_back(0xD9);                                    // EOI
        return RT_OK;
    } catch (...) { return RT_ERR_INVALID; }
}

}  // namespace

extern "C" int rt_encode_jpeg(const uint8_t* rgb, int width, int height, int quality, const char* comment, uint8_t* out,
                              int64_t out_cap, int64_t* out_size) {
    if (!out_size) return RT_ERR_INVALID;
    std::vector<uint8_t> o;
    const int rc = encode(rgb, width, height, quality, comment, o);
    if (rc != RT_OK) return rc;
    *out_size = (int64_t)o.size();
    if (out) {
        if (out_cap < (int64_t)o.size()) return RT_ERR_INVALID;
        memcpy(out, o.data(), o.size());
    }
    return RT_OK;
}

extern "C" int rt_save_jpeg(const char* path, const uint8_t* rgb, int width, int height, int quality, const char* comment) {
    if (!path) return RT_ERR_INVALID;
    std::vector<uint8_t> o;
    const int rc = encode(rgb, width, height, quality, comment, o);
    if (rc != RT_OK) return rc;
    FILE* f = fopen(path, "wb");
    if (!f) return RT_ERR_INVALID;
    const bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
    return (fclose(f) == 0 && ok) ? RT_OK : RT_ERR_INVALID;
}
