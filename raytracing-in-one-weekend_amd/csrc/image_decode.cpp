// JPEG (sequential and progressive, Huffman-coded) and PNG (plain and Adam7-interlaced) decoding for texture files:
// the part of stb_image (v2.27, the copy the
// reference vendors) that Texture's constructor relies on (R/src/Material/Texture.h:11 calls
// stbi_loadf).  Written from the file formats (ITU T.81, RFC 1950/1951, the PNG specification); where
// a format leaves the arithmetic to the decoder -- the inverse DCT, chroma upsampling and the
// YCbCr -> RGB map of JPEG, the 16 -> 8 bit and low-bit-depth conversions of PNG -- the integer
// arithmetic follows stb_image's scalar code paths expression by expression, so that the bytes
// equal stbi_load()'s (tests/test_image_decode.py checks them against fixtures made with the
// reference's own stb_image).  R/ = /root/reference/Raytracing-In-A-Weekend/.
#include "image_decode.h"

#include <cstring>

namespace rt {

namespace {

// ------------------------------------------------------------------------------------ JPEG
// zig-zag position -> natural (row-major) position inside an 8x8 block (ITU T.81 figure A.6)
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {  // canonical code, decoded length by length (T.81 annex C / F.2.2.3)
    int count[17] = {0};
    uint8_t symbol[256] = {0};
    int mincode[17] = {0}, maxcode[18] = {0}, valptr[17] = {0};
    bool present = false;
    bool build() {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k;
            mincode[len] = code;
            code += count[len];
            k += count[len];
            maxcode[len] = count[len] ? code - 1 : -1;
            if (code > (1 << len)) return false;
            code <<= 1;
        }
        return k <= 256;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;  // size in samples, size of the padded plane
    int pred = 0;
    std::vector<uint8_t> plane;
    std::vector<int16_t> coef;  // progressive files: the coefficients of every block of the padded plane, natural order
};

struct JpegReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t bits = 0;
    int nbits = 0;
    int marker = -1;  // a marker met inside entropy-coded data
    bool getByte(int* out) {
        if (p >= end) return false;
        *out = *p++;
        return true;
    }
    void resetBits() {
        bits = 0;
        nbits = 0;
        marker = -1;
    }
    // one bit of the entropy-coded segment; FF00 is a stuffed FF, any other FFxx ends the segment
    // (zero bits are supplied from there on, as decoders conventionally do)
    int bit() {
        if (nbits == 0) {
            int b = 0;
            if (marker < 0 && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    int c = p < end ? *p++ : 0xD9;
                    while (c == 0xFF && p < end) c = *p++;
                    if (c != 0) {
                        marker = c;
                        b = 0;
                    }
                }
            }
            bits = (uint32_t)b;
            nbits = 8;
        }
        nbits--;
        return (int)((bits >> nbits) & 1u);
    }
    int receive(int n) {
        int v = 0;
        for (int i = 0; i < n; i++) v = (v << 1) | bit();
        return v;
    }
    // T.81 F.2.2.1 EXTEND
    int receiveExtend(int n) {
        if (n == 0) return 0;
        int v = receive(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    int decode(const Huffman& h) {
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (code << 1) | bit();
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len])
                return h.symbol[h.valptr[len] + code - h.mincode[len]];
        }
        return -1;
    }
};

inline uint8_t clamp255(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// One pass of the 8-point inverse DCT in 12-bit fixed point (the "slow integer" factorisation of
// the Independent JPEG Group's jidctint, as stb_image uses it: stb_image.h:2404-2443).  Inputs
// s0..s7, outputs the four even-part sums x0..x3 and the four odd-part terms t0..t3 such that the
// eight results are x0+t3, x1+t2, x2+t1, x3+t0, x3-t0, x2-t1, x1-t2, x0-t3.
inline int fix(double x) { return (int)(x * 4096 + 0.5); }
struct Idct1D {
    int x0, x1, x2, x3, t0, t1, t2, t3;
    Idct1D(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
        static const int c0541 = fix(0.5411961f), c1847 = fix(-1.847759065f), c0765 = fix(0.765366865f),
                         c1175 = fix(1.175875602f), c0298 = fix(0.298631336f), c2053 = fix(2.053119869f),
                         c3072 = fix(3.072711026f), c1501 = fix(1.501321110f), c0899 = fix(-0.899976223f),
                         c2562 = fix(-2.562915447f), c1961 = fix(-1.961570560f), c0390 = fix(-0.390180644f);
        int p2 = s2, p3 = s6;
        int p1 = (p2 + p3) * c0541;
        int e2 = p1 + p3 * c1847, e3 = p1 + p2 * c0765;
        int e0 = (s0 + s4) * 4096, e1 = (s0 - s4) * 4096;
        x0 = e0 + e3;
        x3 = e0 - e3;
        x1 = e1 + e2;
        x2 = e1 - e2;
        int o0 = s7, o1 = s5, o2 = s3, o3 = s1;
        int q3 = o0 + o2, q4 = o1 + o3, q1 = o0 + o3, q2 = o1 + o2;
        int q5 = (q3 + q4) * c1175;
        o0 = o0 * c0298;
        o1 = o1 * c2053;
        o2 = o2 * c3072;
        o3 = o3 * c1501;
        q1 = q5 + q1 * c0899;
        q2 = q5 + q2 * c2562;
        q3 = q3 * c1961;
        q4 = q4 * c0390;
        t3 = o3 + q1 + q4;
        t2 = o2 + q2 + q3;
        t1 = o1 + q2 + q4;
        t0 = o0 + q1 + q3;
    }
};

// 8x8 inverse DCT of dequantised coefficients to 8-bit samples: columns first keeping two extra
// bits, then rows with the level shift (+128) folded into the rounding constant
// (stb_image.h:2445-2503).
void idctBlock(uint8_t* out, int stride, const int16_t* d) {
    int v[64];
    for (int i = 0; i < 8; i++) {
        Idct1D c(d[i], d[8 + i], d[16 + i], d[24 + i], d[32 + i], d[40 + i], d[48 + i], d[56 + i]);
        const int x0 = c.x0 + 512, x1 = c.x1 + 512, x2 = c.x2 + 512, x3 = c.x3 + 512;
        v[i] = (x0 + c.t3) >> 10;
        v[56 + i] = (x0 - c.t3) >> 10;
        v[8 + i] = (x1 + c.t2) >> 10;
        v[48 + i] = (x1 - c.t2) >> 10;
        v[16 + i] = (x2 + c.t1) >> 10;
        v[40 + i] = (x2 - c.t1) >> 10;
        v[24 + i] = (x3 + c.t0) >> 10;
        v[32 + i] = (x3 - c.t0) >> 10;
    }
    for (int i = 0; i < 8; i++) {
        const int* r = v + 8 * i;
        uint8_t* o = out + (size_t)i * stride;
        Idct1D c(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
        const int bias = 65536 + (128 << 17);
        const int x0 = c.x0 + bias, x1 = c.x1 + bias, x2 = c.x2 + bias, x3 = c.x3 + bias;
        o[0] = clamp255((x0 + c.t3) >> 17);
        o[7] = clamp255((x0 - c.t3) >> 17);
        o[1] = clamp255((x1 + c.t2) >> 17);
        o[6] = clamp255((x1 - c.t2) >> 17);
        o[2] = clamp255((x2 + c.t1) >> 17);
        o[5] = clamp255((x2 - c.t1) >> 17);
        o[3] = clamp255((x3 + c.t0) >> 17);
        o[4] = clamp255((x3 - c.t0) >> 17);
    }
}

// Chroma upsampling of one output row (stb_image.h:3424-3495, 3614-3623): `near` is the source row
// the output row lies in, `far` its vertical neighbour on the other side.  Returns the row to use.
const uint8_t* upsampleRow(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) return near;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
        return out;
    }
    if (hs == 2 && vs == 1) {
        if (w == 1) {
            out[0] = out[1] = near[0];
            return out;
        }
        out[0] = near[0];
        out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; i++) {
            const int n = 3 * near[i] + 2;
            out[2 * i] = (uint8_t)((n + near[i - 1]) >> 2);
            out[2 * i + 1] = (uint8_t)((n + near[i + 1]) >> 2);
        }
        out[2 * i] = (uint8_t)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
        out[2 * i + 1] = near[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) {
            out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2);
            return out;
        }
        int t1 = 3 * near[0] + far[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) {
            const int t0 = t1;
            t1 = 3 * near[i] + far[i];
            out[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; i++)  // any other factor: nearest neighbour
        for (int j = 0; j < hs; j++) out[i * hs + j] = near[i];
    return out;
}

// YCbCr -> RGB in 20-bit fixed point with the coefficients rounded to 12 bits first
// (stb_image.h:3627-3651).
inline int fix20(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }
void yccToRgb(uint8_t* out, const uint8_t* y, const uint8_t* cbp, const uint8_t* crp, int count) {
    static const int kR = fix20(1.40200f), kGr = fix20(0.71414f), kGb = fix20(0.34414f), kB = fix20(1.77200f);
    for (int i = 0; i < count; i++) {
        const int yf = (y[i] << 20) + (1 << 19);
        const int cr = crp[i] - 128, cb = cbp[i] - 128;
        int r = yf + cr * kR;
        int g = yf + (cr * -kGr) + (int)((uint32_t)(cb * -kGb) & 0xffff0000u);
        int b = yf + cb * kB;
        out[3 * i] = clamp255(r >> 20);
        out[3 * i + 1] = clamp255(g >> 20);
        out[3 * i + 2] = clamp255(b >> 20);
    }
}

bool decodeJpeg(const uint8_t* data, size_t n, Image* img, std::string* err) {
    auto fail = [&](const char* m) {
        *err = std::string("JPEG: ") + m;
        return false;
    };
    JpegReader rd{data + 2, data + n};
    uint16_t quant[4][64];
    bool quantSet[4] = {false, false, false, false};
    Huffman dc[4], ac[4];
    Component comp[4];
    int nComp = 0, width = 0, height = 0, hMax = 1, vMax = 1, mcuX = 0, mcuY = 0, restart = 0;
    bool jfif = false, haveFrame = false, sawScan = false, progressive = false;
    int adobeTransform = -1;
    auto be16 = [&](int* out) {
        int a, b;
        if (!rd.getByte(&a) || !rd.getByte(&b)) return false;
        *out = (a << 8) | b;
        return true;
    };
    for (;;) {
        int m = 0;
        if (rd.marker >= 0) {
            m = rd.marker;
            rd.marker = -1;
        } else {
            int b;
            if (!rd.getByte(&b)) return fail("no end-of-image marker");
            if (b != 0xFF) continue;  // garbage between segments is skipped
            do {
                if (!rd.getByte(&m)) return fail("truncated");
            } while (m == 0xFF);
            if (m == 0) continue;
        }
        if (m == 0xD9) break;  // EOI
        if (m >= 0xD0 && m <= 0xD7) continue;
        int len;
        if (!be16(&len) || len < 2 || rd.p + (len - 2) > rd.end) return fail("bad segment length");
        const uint8_t* seg = rd.p;
        const uint8_t* segEnd = rd.p + (len - 2);
        if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC))
            return fail("lossless, hierarchical and arithmetic-coded JPEG files are not supported");
        if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            if (haveFrame) return fail("two frame headers");
            progressive = m == 0xC2;
            if (len < 8 || seg[0] != 8) return fail("only 8-bit samples are supported");
            height = (seg[1] << 8) | seg[2];
            width = (seg[3] << 8) | seg[4];
            nComp = seg[5];
            if (width <= 0 || height <= 0) return fail("empty image");
            if (nComp != 1 && nComp != 3 && nComp != 4) return fail("bad component count");
            if (len != 8 + 3 * nComp) return fail("bad frame header");
            for (int i = 0; i < nComp; i++) {
                comp[i].id = seg[6 + 3 * i];
                comp[i].h = seg[7 + 3 * i] >> 4;
                comp[i].v = seg[7 + 3 * i] & 15;
                comp[i].tq = seg[8 + 3 * i];
                if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3)
                    return fail("bad component description");
                hMax = std::max(hMax, comp[i].h);
                vMax = std::max(vMax, comp[i].v);
            }
            for (int i = 0; i < nComp; i++)
                if (hMax % comp[i].h || vMax % comp[i].v) return fail("unsupported sampling factors");
            if ((uint64_t)width * height * nComp > (1ull << 31)) return fail("image too large");
            mcuX = (width + hMax * 8 - 1) / (hMax * 8);
            mcuY = (height + vMax * 8 - 1) / (vMax * 8);
            for (int i = 0; i < nComp; i++) {
                Component& c = comp[i];
                c.x = (width * c.h + hMax - 1) / hMax;
                c.y = (height * c.v + vMax - 1) / vMax;
                c.w2 = mcuX * c.h * 8;
                c.h2 = mcuY * c.v * 8;
                c.plane.assign((size_t)c.w2 * c.h2, 0);
                if (progressive) c.coef.assign((size_t)c.w2 * c.h2, 0);
            }
            haveFrame = true;
        } else if (m == 0xC4) {  // Huffman tables
            const uint8_t* q = seg;
            while (q < segEnd) {
                if (q + 17 > segEnd) return fail("bad Huffman table");
                const int tc = q[0] >> 4, th = q[0] & 15;
                if (tc > 1 || th > 3) return fail("bad Huffman table id");
                Huffman& h = tc ? ac[th] : dc[th];
                int total = 0;
                for (int i = 1; i <= 16; i++) {
                    h.count[i] = q[i];
                    total += q[i];
                }
                q += 17;
                if (total > 256 || q + total > segEnd) return fail("bad Huffman table");
                std::memcpy(h.symbol, q, (size_t)total);
                q += total;
                if (!h.build()) return fail("bad Huffman code lengths");
                h.present = true;
            }
        } else if (m == 0xDB) {  // quantisation tables, stored in natural order
            const uint8_t* q = seg;
            while (q < segEnd) {
                const int pq = q[0] >> 4, tq = q[0] & 15;
                if (pq > 1 || tq > 3) return fail("bad quantisation table");
                q++;
                if (q + (pq ? 128 : 64) > segEnd) return fail("bad quantisation table");
                for (int i = 0; i < 64; i++) {
                    quant[tq][kZigzag[i]] = (uint16_t)(pq ? ((q[0] << 8) | q[1]) : q[0]);
                    q += pq ? 2 : 1;
                }
                quantSet[tq] = true;
            }
        } else if (m == 0xDD) {
            if (len != 4) return fail("bad restart interval");
            restart = (seg[0] << 8) | seg[1];
        } else if (m == 0xE0) {
            if (len >= 7 && std::memcmp(seg, "JFIF\0", 5) == 0) jfif = true;
        } else if (m == 0xEE) {
            if (len >= 14 && std::memcmp(seg, "Adobe\0", 6) == 0) adobeTransform = seg[11];
        } else if (m == 0xDA) {  // scan
            if (!haveFrame) return fail("scan before frame header");
            if (len < 3) return fail("bad scan header");  // seg[0] lies inside the segment from here on
            const int ns = seg[0];
            if (ns < 1 || ns > nComp || len != 6 + 2 * ns) return fail("bad scan header");
            int order[4];
            for (int i = 0; i < ns; i++) {
                int which = -1;
                for (int k = 0; k < nComp; k++)
                    if (comp[k].id == seg[1 + 2 * i]) which = k;
                if (which < 0) return fail("scan names an unknown component");
                comp[which].td = seg[2 + 2 * i] >> 4;
                comp[which].ta = seg[2 + 2 * i] & 15;
                if (comp[which].td > 3 || comp[which].ta > 3) return fail("bad table selector");
                order[i] = which;
            }
            // spectral selection and successive approximation (T.81 annex G; stb_image.h:3074-3087)
            const int specStart = seg[1 + 2 * ns], specEnd = seg[2 + 2 * ns], succHigh = seg[3 + 2 * ns] >> 4, succLow = seg[3 + 2 * ns] & 15;
            if (progressive) {
                if (specStart > 63 || specEnd > 63 || specStart > specEnd || succHigh > 13 || succLow > 13) return fail("bad scan header");
                if (specStart == 0 && specEnd != 0) return fail("a scan with DC and AC coefficients in a progressive file");
                if (specStart != 0 && ns != 1) return fail("an interleaved AC scan in a progressive file");
            } else if (specStart != 0 || specEnd != 63 || succHigh != 0 || succLow != 0) {
                return fail("not a sequential scan");
            }
            rd.p = segEnd;
            rd.resetBits();
            for (int k = 0; k < nComp; k++) comp[k].pred = 0;
            int eobRun = 0;
            // One block of a progressive scan (T.81 G.1.2, G.2; stb_image.h:2185-2323): the DC coefficient's first
            // scan or one more of its bits; a band of AC coefficients' first scan (with runs of blocks that end
            // early), or one more bit of the band (new coefficients come in as +-1 << succLow between the
            // correction bits of the ones that are already non-zero).
            auto blockProgressive = [&](Component& c, int bx, int by) -> bool {
                int16_t* d = &c.coef[((size_t)by * (c.w2 / 8) + (size_t)bx) * 64];
                if (specStart == 0) {
                    if (succHigh == 0) {
                        const Huffman& hd = dc[c.td];
                        if (!hd.present) return false;
                        std::memset(d, 0, 64 * sizeof(int16_t));
                        const int t = rd.decode(hd);
                        if (t < 0 || t > 15) return false;
                        c.pred += rd.receiveExtend(t);
                        d[0] = (int16_t)(c.pred * (1 << succLow));
                    } else if (rd.bit()) {
                        d[0] = (int16_t)(d[0] + (int16_t)(1 << succLow));
                    }
                    return true;
                }
                const Huffman& ha = ac[c.ta];
                if (!ha.present) return false;
                if (succHigh == 0) {
                    if (eobRun) {
                        --eobRun;
                        return true;
                    }
                    int k = specStart;
                    do {
                        const int rs = rd.decode(ha);
                        if (rs < 0) return false;
                        const int sBits = rs & 15, r = rs >> 4;
                        if (sBits == 0) {
                            if (r < 15) {
                                eobRun = 1 << r;
                                if (r) eobRun += rd.receive(r);
                                --eobRun;
                                break;
                            }
                            k += 16;
                        } else {
                            k += r;
                            if (k > 63) return false;
                            d[kZigzag[k++]] = (int16_t)(rd.receiveExtend(sBits) * (1 << succLow));
                        }
                    } while (k <= specEnd);
                    return true;
                }
                const int16_t bit = (int16_t)(1 << succLow);
                auto correct = [&](int16_t* q) {  // one more bit of a coefficient that is already non-zero
                    if (rd.bit() && (*q & bit) == 0) *q = (int16_t)(*q > 0 ? *q + bit : *q - bit);
                };
                if (eobRun) {
                    --eobRun;
                    for (int k = specStart; k <= specEnd; k++) {
                        int16_t* q = &d[kZigzag[k]];
                        if (*q != 0) correct(q);
                    }
                    return true;
                }
                int k = specStart;
                do {
                    const int rs = rd.decode(ha);
                    if (rs < 0) return false;
                    int sVal = rs & 15, r = rs >> 4;
                    if (sVal == 0) {
                        if (r < 15) {
                            eobRun = (1 << r) - 1;
                            if (r) eobRun += rd.receive(r);
                            r = 64;  // to the end of the band: only correction bits from here
                        }
                        // (r == 15: sixteen zero coefficients -- fifteen skipped, the sixteenth "written" as 0)
                    } else {
                        if (sVal != 1) return false;
                        sVal = rd.bit() ? bit : -bit;
                    }
                    while (k <= specEnd) {
                        int16_t* q = &d[kZigzag[k++]];
                        if (*q != 0) {
                            correct(q);
                        } else {
                            if (r == 0) {
                                *q = (int16_t)sVal;
                                break;
                            }
                            --r;
                        }
                    }
                } while (k <= specEnd);
                return true;
            };
            auto block = [&](Component& c, int bx, int by) -> bool {
                if (progressive) return blockProgressive(c, bx, by);
                const Huffman &hd = dc[c.td], &ha = ac[c.ta];
                if (!hd.present || !ha.present || !quantSet[c.tq]) return false;
                const uint16_t* q = quant[c.tq];
                int16_t coef[64];
                std::memset(coef, 0, sizeof coef);
                const int t = rd.decode(hd);
                if (t < 0 || t > 15) return false;
                c.pred += rd.receiveExtend(t);
                coef[0] = (int16_t)(c.pred * q[0]);
                for (int k = 1; k < 64;) {
                    const int rs = rd.decode(ha);
                    if (rs < 0) return false;
                    const int s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (rs != 0xF0) break;  // end of block
                        k += 16;
                    } else {
                        k += r;
                        if (k > 63) return false;
                        const int z = kZigzag[k++];
                        coef[z] = (int16_t)(rd.receiveExtend(s) * q[z]);
                    }
                }
                idctBlock(&c.plane[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2, coef);
                return true;
            };
            int todo = restart ? restart : 0x7fffffff;
            auto afterUnit = [&]() -> bool {  // restart markers reset the predictors
                if (--todo <= 0) {
                    rd.nbits = 0;
                    if (rd.marker < 0) {  // the marker follows byte-aligned
                        int a, b;
                        if (rd.getByte(&a) && a == 0xFF && rd.getByte(&b)) rd.marker = b;
                    }
                    if (rd.marker >= 0xD0 && rd.marker <= 0xD7) {
                        rd.resetBits();
                        for (int k = 0; k < nComp; k++) comp[k].pred = 0;
                        eobRun = 0;
                        todo = restart;
                    } else {
                        return false;  // no restart marker: the scan is over
                    }
                }
                return true;
            };
            bool more = true;
            if (ns == 1) {  // non-interleaved: the component's own blocks, row by row
                Component& c = comp[order[0]];
                const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
                for (int by = 0; by < bh && more; by++)
                    for (int bx = 0; bx < bw && more; bx++) {
                        if (!block(c, bx, by)) return fail("corrupt entropy-coded data");
                        more = afterUnit();
                    }
            } else {
                for (int my = 0; my < mcuY && more; my++)
                    for (int mx = 0; mx < mcuX && more; mx++) {
                        for (int i = 0; i < ns; i++) {
                            Component& c = comp[order[i]];
                            for (int y = 0; y < c.v; y++)
                                for (int x = 0; x < c.h; x++)
                                    if (!block(c, mx * c.h + x, my * c.v + y)) return fail("corrupt entropy-coded data");
                        }
                        more = afterUnit();
                    }
            }
            sawScan = true;
            continue;  // rd.p stands after the entropy-coded data (or rd.marker holds the next marker)
        }
        rd.p = segEnd;
    }
    if (!haveFrame || !sawScan) return fail("no image data");
    if (progressive) {
        // every scan has been read: dequantise and transform the blocks that hold image samples (stb_image.h:3107-3125)
        for (int i = 0; i < nComp; i++) {
            Component& c = comp[i];
            if (!quantSet[c.tq]) return fail("missing quantisation table");
            const uint16_t* q = quant[c.tq];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    int16_t* d = &c.coef[((size_t)by * (c.w2 / 8) + (size_t)bx) * 64];
                    for (int k = 0; k < 64; k++) d[k] = (int16_t)(d[k] * q[k]);
                    idctBlock(&c.plane[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2, d);
                }
        }
    }

    // ---- upsample and colour-convert, row by row (stb_image.h:3858-3996)
    const int outN = nComp >= 3 ? 3 : 1;
    img->w = width;
    img->h = height;
    img->channels = outN;
    img->px.assign((size_t)width * height * outN, 0);
    const bool isRgb = nComp == 3 && ((comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') ||
                                      (adobeTransform == 0 && !jfif));
    struct Up {
        int hs, vs, wLo, ystep, ypos;
        const uint8_t *line0, *line1;
        std::vector<uint8_t> buf;
    } up[4];
    for (int k = 0; k < nComp; k++) {
        up[k].hs = hMax / comp[k].h;
        up[k].vs = vMax / comp[k].v;
        up[k].ystep = up[k].vs >> 1;
        up[k].wLo = (width + up[k].hs - 1) / up[k].hs;
        up[k].ypos = 0;
        up[k].line0 = up[k].line1 = comp[k].plane.data();
        up[k].buf.assign((size_t)width + 8, 0);
    }
    for (int j = 0; j < height; j++) {
        const uint8_t* row[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < nComp; k++) {
            Up& u = up[k];
            const bool bottom = u.ystep >= (u.vs >> 1);
            row[k] = upsampleRow(u.buf.data(), bottom ? u.line1 : u.line0, bottom ? u.line0 : u.line1, u.wLo, u.hs, u.vs);
            if (++u.ystep >= u.vs) {
                u.ystep = 0;
                u.line0 = u.line1;
                if (++u.ypos < comp[k].y) u.line1 += comp[k].w2;
            }
        }
        uint8_t* out = &img->px[(size_t)j * width * outN];
        if (nComp == 1) {
            std::memcpy(out, row[0], (size_t)width);
        } else if (nComp == 4) {
            // four components (stb_image.h:3924-3947): CMYK as Adobe writes it (stored inverted: a product with K),
            // YCCK (the first three through the YCbCr map, then inverted and multiplied with K), or -- without an
            // Adobe marker that says which -- YCbCr with a fourth channel that is ignored
            auto blinn = [](int x, int y) {  // x * y / 255, rounded (stb_image.h:3718-3722)
                const unsigned t = (unsigned)(x * y + 128);
                return (uint8_t)((t + (t >> 8)) >> 8);
            };
            if (adobeTransform == 0) {
                for (int i = 0; i < width; i++)
                    for (int k = 0; k < 3; k++) out[3 * i + k] = blinn(row[k][i], row[3][i]);
            } else {
                yccToRgb(out, row[0], row[1], row[2], width);
                if (adobeTransform == 2)
                    for (int i = 0; i < width; i++)
                        for (int k = 0; k < 3; k++) out[3 * i + k] = blinn(255 - out[3 * i + k], row[3][i]);
            }
        } else if (isRgb) {
            for (int i = 0; i < width; i++) {
                out[3 * i] = row[0][i];
                out[3 * i + 1] = row[1][i];
                out[3 * i + 2] = row[2][i];
            }
        } else {
            yccToRgb(out, row[0], row[1], row[2], width);
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------ PNG
struct BitsLsb {  // DEFLATE packs bits starting at the least significant bit of each byte
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int n = 0;
    bool ok = true;
    int get(int count) {
        while (n < count) {
            if (p >= end) {
                ok = false;
                return 0;
            }
            acc |= (uint32_t)(*p++) << n;
            n += 8;
        }
        const int v = (int)(acc & ((1u << count) - 1u));
        acc >>= count;
        n -= count;
        return v;
    }
};

struct DeflateCode {  // canonical Huffman code of RFC 1951, decoded bit by bit
    uint16_t count[16] = {0};
    uint16_t symbol[288] = {0};
    bool build(const uint8_t* lengths, int n) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for (int len = 1; len < 16; len++) {
            left = (left << 1) - count[len];
            if (left < 0) return false;  // over-subscribed
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
        for (int i = 0; i < n; i++)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitsLsb& b) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; len++) {
            code |= b.get(1);
            if (!b.ok) return -1;
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

bool inflateZlib(const std::vector<uint8_t>& z, std::vector<uint8_t>* out, size_t expect, std::string* err) {
    auto fail = [&](const char* m) {
        *err = std::string("PNG: ") + m;
        return false;
    };
    if (z.size() < 2 || (z[0] & 15) != 8 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 32)) return fail("bad zlib header");
    BitsLsb b{z.data() + 2, z.data() + z.size()};
    out->clear();
    out->reserve(expect);
    // A few bytes of deflate data can expand a thousandfold: output is cut off at twice what the image header
    // announces (plus one window), long before it can exhaust memory.  (stb_image inflates whatever the stream
    // holds and ignores the surplus; files that far off their own header are refused here instead.)
    const size_t ceiling = 2 * expect + 65536;
    auto tooMuch = [&]() { return out->size() > ceiling; };
    static const uint16_t lenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t distBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t distExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        const int last = b.get(1), type = b.get(2);
        if (!b.ok) return fail("truncated zlib stream");
        if (type == 0) {
            b.acc = 0;
            b.n = 0;  // to the byte boundary
            if (b.p + 4 > b.end) return fail("truncated stored block");
            const int len = b.p[0] | (b.p[1] << 8), nlen = b.p[2] | (b.p[3] << 8);
            if ((len ^ 0xFFFF) != nlen) return fail("corrupt stored block");
            b.p += 4;
            if (b.p + len > b.end) return fail("truncated stored block");
            out->insert(out->end(), b.p, b.p + len);
            b.p += len;
            if (tooMuch()) return fail("far more image data than the header announces");
        } else if (type == 1 || type == 2) {
            DeflateCode lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                for (int i = 0; i < 288; i++) lengths[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
                lit.build(lengths, 288);
                for (int i = 0; i < 30; i++) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                const int nlen = b.get(5) + 257, ndist = b.get(5) + 1, ncode = b.get(4) + 4;
                if (!b.ok || nlen > 286 || ndist > 30) return fail("bad dynamic block header");
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)b.get(3);
                DeflateCode clc;
                if (!clc.build(cl, 19)) return fail("bad code-length code");
                int i = 0;
                while (i < nlen + ndist) {
                    const int sym = clc.decode(b);
                    if (sym < 0) return fail("corrupt code lengths");
                    if (sym < 16) {
                        lengths[i++] = (uint8_t)sym;
                    } else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (i == 0) return fail("corrupt code lengths");
                            val = lengths[i - 1];
                            rep = 3 + b.get(2);
                        } else if (sym == 17) {
                            rep = 3 + b.get(3);
                        } else {
                            rep = 11 + b.get(7);
                        }
                        if (i + rep > nlen + ndist) return fail("corrupt code lengths");
                        while (rep--) lengths[i++] = (uint8_t)val;
                    }
                }
                if (!b.ok || !lit.build(lengths, nlen) || !dist.build(lengths + nlen, ndist)) return fail("bad Huffman codes");
            }
            for (;;) {
                const int sym = lit.decode(b);
                if (sym < 0) return fail("corrupt compressed data");
                if (sym < 256) {
                    out->push_back((uint8_t)sym);
                } else if (sym == 256) {
                    break;
                } else {
                    if (sym > 285) return fail("corrupt compressed data");
                    const int len = lenBase[sym - 257] + b.get(lenExtra[sym - 257]);
                    const int ds = dist.decode(b);
                    if (ds < 0 || ds > 29) return fail("corrupt compressed data");
                    const size_t d = (size_t)distBase[ds] + (size_t)b.get(distExtra[ds]);
                    if (!b.ok || d > out->size()) return fail("corrupt compressed data");
                    const size_t start = out->size() - d;
                    for (int i = 0; i < len; i++) out->push_back((*out)[start + (size_t)i]);
                    if (tooMuch()) return fail("far more image data than the header announces");
                }
            }
        } else {
            return fail("bad block type");
        }
        if (last) break;
    }
    return true;
}

bool decodePng(const uint8_t* data, size_t n, Image* img, std::string* err) {
    auto fail = [&](const char* m) {
        *err = std::string("PNG: ") + m;
        return false;
    };
    size_t pos = 8;
    int width = 0, height = 0, depth = 0, color = -1, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool header = false, done = false;
    while (!done) {
        if (pos + 8 > n) return fail("truncated");
        const uint32_t len = ((uint32_t)data[pos] << 24) | (data[pos + 1] << 16) | (data[pos + 2] << 8) | data[pos + 3];
        const uint8_t* type = data + pos + 4;
        const uint8_t* body = data + pos + 8;
        if ((uint64_t)pos + 12 + len > n) return fail("truncated chunk");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return fail("bad IHDR");
            width = (int)(((uint32_t)body[0] << 24) | (body[1] << 16) | (body[2] << 8) | body[3]);
            height = (int)(((uint32_t)body[4] << 24) | (body[5] << 16) | (body[6] << 8) | body[7]);
            depth = body[8];
            color = body[9];
            interlace = body[12];
            if (width <= 0 || height <= 0 || (uint64_t)width * height > (1ull << 28)) return fail("bad size");
            if (body[10] || body[11]) return fail("unknown compression or filter method");
            if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) return fail("bad bit depth");
            if (color != 0 && color != 2 && color != 3 && color != 4 && color != 6) return fail("bad colour type");
            if ((color == 3 && depth == 16) || ((color == 2 || color == 4 || color == 6) && depth < 8)) return fail("bad bit depth");
            if (interlace > 1) return fail("unknown interlace method");
            header = true;
        } else if (!header) {
            return fail("first chunk is not IHDR");
        } else if (!std::memcmp(type, "PLTE", 4)) {
            if (len > 768 || len % 3) return fail("bad PLTE");
            palette.assign(body, body + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            done = true;
        }
        pos += 12 + (size_t)len;
    }
    const int fileCh = color == 0 ? 1 : (color == 2 ? 3 : (color == 3 ? 1 : (color == 4 ? 2 : 4)));
    const int bpp = std::max(1, fileCh * depth / 8);  // distance to the "previous pixel" of the filters
    auto rowBytesOf = [&](int w) { return ((size_t)w * fileCh * depth + 7) / 8; };
    // The image data is one filtered image, or with Adam7 interlacing seven of them one after the other: pass p
    // holds the pixels (xorig + i * xspc, yorig + j * yspc) (PNG specification, section 8.2; stb_image.h:4740-4772).
    struct Pass {
        int x0, y0, dx, dy, w, h;
    };
    std::vector<Pass> passes;
    if (interlace) {
        static const int xorig[7] = {0, 4, 0, 2, 0, 1, 0}, yorig[7] = {0, 0, 4, 0, 2, 0, 1};
        static const int xspc[7] = {8, 8, 4, 4, 2, 2, 1}, yspc[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int p = 0; p < 7; p++) {
            const int w = (width - xorig[p] + xspc[p] - 1) / xspc[p], h = (height - yorig[p] + yspc[p] - 1) / yspc[p];
            if (w > 0 && h > 0) passes.push_back({xorig[p], yorig[p], xspc[p], yspc[p], w, h});
        }
    } else {
        passes.push_back({0, 0, 1, 1, width, height});
    }
    size_t expect = 0;
    for (const Pass& ps : passes) expect += (rowBytesOf(ps.w) + 1) * (size_t)ps.h;
    std::vector<uint8_t> raw;
    if (!inflateZlib(idat, &raw, expect, err)) return false;
    if (raw.size() < expect) return fail("not enough image data");
    const size_t nSamples = (size_t)width * height * fileCh;
    std::vector<uint8_t> s8(nSamples);
    std::vector<uint16_t> s16;
    if (depth == 16) s16.resize(nSamples);
    size_t at = 0;
    std::vector<uint8_t> pix;
    for (const Pass& ps : passes) {
        const size_t rowBytes = rowBytesOf(ps.w);
        // ---- undo the row filters (PNG specification, section 9)
        pix.assign(rowBytes * (size_t)ps.h, 0);
        for (int y = 0; y < ps.h; y++) {
            const uint8_t* src = &raw[at + (rowBytes + 1) * (size_t)y];
            uint8_t* cur = &pix[rowBytes * (size_t)y];
            const uint8_t* up = y ? cur - rowBytes : nullptr;
            const int f = *src++;
            if (f > 4) return fail("bad filter type");
            for (size_t i = 0; i < rowBytes; i++) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
                int pred = 0;
                if (f == 1) pred = a;
                else if (f == 2) pred = b;
                else if (f == 3) pred = (a + b) >> 1;
                else if (f == 4) {
                    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                }
                cur[i] = (uint8_t)(src[i] + pred);
            }
        }
        at += (rowBytes + 1) * (size_t)ps.h;
        // ---- to 8-bit samples at the pass's pixels: low bit depths are unpacked (grey levels scaled to 0..255,
        // palette indices kept), 16-bit samples keep their high byte (stb_image.h:4774-4801, 1208-1222)
        const int scale = (depth < 8 && color == 0) ? (depth == 1 ? 0xff : (depth == 2 ? 0x55 : 0x11)) : 1;
        for (int y = 0; y < ps.h; y++)
            for (int x = 0; x < ps.w; x++) {
                const size_t dst = ((size_t)(ps.y0 + y * ps.dy) * width + (size_t)(ps.x0 + x * ps.dx)) * fileCh;
                for (int k = 0; k < fileCh; k++) {
                    const size_t i = (size_t)x * fileCh + k;
                    const uint8_t* row = &pix[rowBytes * (size_t)y];
                    if (depth == 8) {
                        s8[dst + k] = row[i];
                    } else if (depth == 16) {
                        s16[dst + k] = (uint16_t)((row[2 * i] << 8) | row[2 * i + 1]);
                        s8[dst + k] = row[2 * i];
                    } else {
                        const size_t bit = i * depth;
                        const int v = (row[bit / 8] >> (8 - depth - (int)(bit % 8))) & ((1 << depth) - 1);
                        s8[dst + k] = (uint8_t)(v * scale);
                    }
                }
            }
    }
    // ---- palette expansion / colour-key transparency (stb_image.h:4803-4880, 5050-5100)
    const size_t npx = (size_t)width * height;
    if (color == 3) {
        if (palette.empty()) return fail("palette image without PLTE");
        const bool alpha = !trns.empty();
        img->channels = alpha ? 4 : 3;
        img->px.resize(npx * img->channels);
        uint8_t pal[256][4];
        for (int i = 0; i < 256; i++) {
            const bool in = (size_t)i * 3 + 2 < palette.size();
            pal[i][0] = in ? palette[3 * i] : 0;
            pal[i][1] = in ? palette[3 * i + 1] : 0;
            pal[i][2] = in ? palette[3 * i + 2] : 0;
            pal[i][3] = (size_t)i < trns.size() ? trns[i] : 255;
        }
        for (size_t i = 0; i < npx; i++) std::memcpy(&img->px[i * img->channels], pal[s8[i]], (size_t)img->channels);
    } else if (!trns.empty() && (color == 0 || color == 2)) {
        if (trns.size() < (size_t)fileCh * 2) return fail("bad tRNS");
        img->channels = fileCh + 1;
        img->px.resize(npx * img->channels);
        uint16_t key16[3];
        uint8_t key8[3];
        const int scale = (color == 0 && depth < 8) ? (depth == 1 ? 0xff : (depth == 2 ? 0x55 : 0x11)) : 1;
        for (int k = 0; k < fileCh; k++) {
            key16[k] = (uint16_t)((trns[2 * k] << 8) | trns[2 * k + 1]);
            key8[k] = (uint8_t)((key16[k] & 255) * scale);
        }
        for (size_t i = 0; i < npx; i++) {
            bool match = true;
            for (int k = 0; k < fileCh; k++)
                match = match && (depth == 16 ? s16[i * fileCh + k] == key16[k] : s8[i * fileCh + k] == key8[k]);
            for (int k = 0; k < fileCh; k++) img->px[i * img->channels + k] = s8[i * fileCh + k];
            img->px[i * img->channels + fileCh] = match ? 0 : 255;
        }
    } else {
        img->channels = fileCh;
        img->px = s8;
    }
    img->w = width;
    img->h = height;
    return true;
}

}  // namespace

bool decodeImage(const uint8_t* data, size_t n, Image* out, std::string* err) {
    static const uint8_t pngSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n >= 8 && std::memcmp(data, pngSig, 8) == 0) return decodePng(data, n, out, err);
    if (n >= 4 && data[0] == 0xFF && data[1] == 0xD8) return decodeJpeg(data, n, out, err);
    *err = "not a JPEG or PNG file";
    return false;
}

// What Texture's constructor keeps: three channels per texel.  The reference copies the first
// three floats per texel whatever the channel count (R/src/Material/Texture.h:17-22), which is only
// meaningful for 3-channel files; grey images are replicated and alpha is dropped here instead.
void toRgb(const Image& in, std::vector<uint8_t>* rgb) {
    const size_t npx = (size_t)in.w * in.h;
    rgb->resize(npx * 3);
    for (size_t i = 0; i < npx; i++) {
        const uint8_t* p = &in.px[i * in.channels];
        uint8_t* o = &(*rgb)[3 * i];
        if (in.channels >= 3) {
            o[0] = p[0];
            o[1] = p[1];
            o[2] = p[2];
        } else {
            o[0] = o[1] = o[2] = p[0];
        }
    }
}

}  // namespace rt
