// JPEG -> RGB8 on the host: the role of `ImageReader::open(..).decode().to_rgb8()` (texture.rs:62-67) for the two JPEG files the
// reference's scene scripts open — assets/earthmap.jpg (baseline, scene 2, main.rs:100) and assets/envmap.jpg (progressive,
// 7616x3808, scene 5, main.rs:365). The reference decodes them with the `image` crate's zune-jpeg 0.4.13, which is not in this
// container: the decoder's output is therefore "parity unpinned" against the reference itself. What it IS pinned to is the
// JPEG standard's reference arithmetic as libjpeg defines it — the accurate integer inverse DCT (jidctint.c "ISLOW", 13-bit
// constants), libjpeg's fixed-point YCbCr -> RGB tables (jdcolor.c) and its "fancy" triangle-filter chroma upsampling
// (jdsample.c) — which tests/test_host_api.py checks pixel for pixel against Pillow (libjpeg-turbo) on both assets and on
// synthetic files of every supported layout. Not on the hot path: runs once per texture at scene build.
//
// Supported: 8-bit baseline / extended sequential (SOF0, SOF1) and progressive (SOF2) Huffman JPEG, 1 component (grey) or 3
// (YCbCr, or RGB when an Adobe APP14 marker says so), sampling factors 1 or 2 per axis, restart intervals. Rejected with an
// error: arithmetic coding, lossless, hierarchical, 12-bit, CMYK.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pt_amd.h"
#include "pt_scene.h"

using pt::set_error;

namespace {

struct JpegError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] void bad(const std::string& what) { throw JpegError(what); }

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t bits[17] = {0};   // bits[l] = number of codes of length l
    uint8_t vals[256] = {0};
    int32_t maxcode[18];      // largest code of length l (-1: none)
    int32_t valptr[17];
    int32_t mincode[17];
    uint16_t look[512];       // 9-bit lookahead: (length << 8) | symbol, 0 = longer than 9 bits
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
        maxcode[17] = 0x7FFFFFFF;
        memset(look, 0, sizeof look);
        code = 0;
        k = 0;
        for (int l = 1; l <= 9; ++l) {
            for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                const int first = code << (9 - l);
                for (int j = 0; j < (1 << (9 - l)); ++j) look[first + j] = (uint16_t)((l << 8) | vals[k]);
            }
            code <<= 1;
        }
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;                 // tables of the current scan
    int bw = 0, bh = 0;                 // blocks per row / column of the MCU-padded grid
    int cw = 0, ch = 0;                 // real size in samples: ceil(W * h / hmax), ceil(H * v / vmax)
    int pred = 0;
    std::vector<int16_t> coef;          // bw * bh blocks of 64, natural order
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int n = 0;
    int marker = 0;                     // a marker met inside the entropy-coded data (zeros are fed from then on, as libjpeg does)
    void fill() {
        while (n <= 56) {
            uint32_t b = 0;
            if (!marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    while (p < end && *p == 0xFF) ++p;             // fill bytes
                    const uint32_t m = p < end ? *p++ : 0xD9u;
                    if (m != 0) { marker = (int)m; b = 0; }        // not a stuffed zero: a marker
                }
            }
            acc |= (uint64_t)b << (56 - n);
            n += 8;
        }
    }
    uint32_t peek(int k) {
        if (n < k) fill();
        return (uint32_t)(acc >> (64 - k));
    }
    void skip(int k) { acc <<= k; n -= k; }
    uint32_t get(int k) {
        if (k == 0) return 0;
        const uint32_t v = peek(k);
        skip(k);
        return v;
    }
    void restart() { acc = 0; n = 0; marker = 0; }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }   // F.12

inline int decode(BitReader& br, const Huff& h) {
    const uint32_t v = br.peek(16);
    const uint16_t lk = h.look[v >> 7];
    if (lk) {
        br.skip(lk >> 8);
        return lk & 0xFF;
    }
    int l = 10;
    int32_t code;
    for (;;) {
        code = (int32_t)(v >> (16 - l));
        if (code <= h.maxcode[l]) break;
        if (++l > 16) bad("corrupt Huffman code");
    }
    br.skip(l);
    return h.vals[h.valptr[l] + (code - h.mincode[l])];
}

// ---- jidctint.c "ISLOW": accurate integer inverse DCT, 13-bit constants, two passes with 2 extra bits between them ------------
void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr long F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299, F_1_847 = 15137,
                   F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
    auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
    long ws[64];
    for (int c = 0; c < 8; ++c) {
        auto d = [&](int r) { return (long)in[8 * r + c] * (long)q[8 * r + c]; };
        long z2 = d(2), z3 = d(6);
        long z1 = (z2 + z3) * F_0_541;
        long tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
        z2 = d(0); z3 = d(4);
        long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);      // (libjpeg shifts; a shift of a negative value is UB before C++20)
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = d(7); tmp1 = d(5); tmp2 = d(3); tmp3 = d(1);
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[8 * 0 + c] = descale(tmp10 + tmp3, CB - P1); ws[8 * 7 + c] = descale(tmp10 - tmp3, CB - P1);
        ws[8 * 1 + c] = descale(tmp11 + tmp2, CB - P1); ws[8 * 6 + c] = descale(tmp11 - tmp2, CB - P1);
        ws[8 * 2 + c] = descale(tmp12 + tmp1, CB - P1); ws[8 * 5 + c] = descale(tmp12 - tmp1, CB - P1);
        ws[8 * 3 + c] = descale(tmp13 + tmp0, CB - P1); ws[8 * 4 + c] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; ++r) {
        const long* w = ws + 8 * r;
        long z2 = w[2], z3 = w[6];
        long z1 = (z2 + z3) * F_0_541;
        long tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
        long tmp0 = (w[0] + w[4]) * (1L << CB), tmp1 = (w[0] - w[4]) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        auto put = [&](int c, long x) {
            const long s = descale(x, CB + P1 + 3) + 128;                       // range_limit: centre on 128, clamp to a byte
            out[(size_t)r * stride + c] = (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
        };
        put(0, tmp10 + tmp3); put(7, tmp10 - tmp3); put(1, tmp11 + tmp2); put(6, tmp11 - tmp2);
        put(2, tmp12 + tmp1); put(5, tmp12 - tmp1); put(3, tmp13 + tmp0); put(4, tmp13 - tmp0);
    }
}

struct Decoder {
    const uint8_t* d;
    size_t n, pos = 2;
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool progressive = false, have_sof = false;
    int restart_interval = 0;
    int adobe_transform = -1;
    uint16_t qt[4][64];
    bool have_qt[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    Component comp[3];

    uint32_t be16(size_t p) const {
        if (p + 2 > n) bad("truncated file");
        return ((uint32_t)d[p] << 8) | d[p + 1];
    }

    void read_dqt(size_t p, size_t end) {
        while (p < end) {
            const int pq = d[p] >> 4, tq = d[p] & 15;
            ++p;
            if (tq > 3 || pq > 1) bad("bad quantisation table");
            if (p + (pq ? 128 : 64) > end) bad("truncated quantisation table");
            for (int i = 0; i < 64; ++i) {
                qt[tq][ZIGZAG[i]] = pq ? (uint16_t)(((uint32_t)d[p] << 8) | d[p + 1]) : d[p];
                p += pq ? 2 : 1;
            }
            have_qt[tq] = true;
        }
    }
    void read_dht(size_t p, size_t end) {
        while (p < end) {
            if (p + 17 > end) bad("truncated Huffman table");
            const int tc = d[p] >> 4, th = d[p] & 15;
            if (tc > 1 || th > 3) bad("bad Huffman table id");
            Huff& h = tc ? ac[th] : dc[th];
            int total = 0;
            h.bits[0] = 0;
            for (int l = 1; l <= 16; ++l) { h.bits[l] = d[p + l]; total += h.bits[l]; }
            p += 17;
            if (total > 256 || p + (size_t)total > end) bad("bad Huffman table");
            // a table whose code space overflows 16 bits is malformed
            int code = 0;
            for (int l = 1; l <= 16; ++l) { code += h.bits[l]; if (code > (1 << l)) bad("over-subscribed Huffman table"); code <<= 1; }
            memset(h.vals, 0, sizeof h.vals);
            memcpy(h.vals, d + p, (size_t)total);
            p += (size_t)total;
            h.present = true;
            h.build();
        }
    }
    void read_sof(size_t p, size_t end, int marker) {
        if (have_sof) bad("more than one frame header");
        if (end - p < 6) bad("truncated frame header");
        if (d[p] != 8) bad("only 8-bit JPEG is supported");
        H = (int)be16(p + 1);
        W = (int)be16(p + 3);
        ncomp = d[p + 5];
        if (W <= 0 || H <= 0) bad("bad image size");
        if ((uint64_t)W * (uint64_t)H > (1ull << 28)) bad("image too large");
        if (ncomp != 1 && ncomp != 3) bad("only grey and three-component JPEG are supported");
        if (end - p < 6 + 3 * (size_t)ncomp) bad("truncated frame header");
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            c.id = d[p + 6 + 3 * i];
            c.h = d[p + 7 + 3 * i] >> 4;
            c.v = d[p + 7 + 3 * i] & 15;
            c.tq = d[p + 8 + 3 * i];
            if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) bad("unsupported sampling factors");
            hmax = std::max(hmax, c.h);
            vmax = std::max(vmax, c.v);
        }
        if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }       // a single-component scan is never interleaved
        mcux = (W + 8 * hmax - 1) / (8 * hmax);
        mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            if (hmax % c.h || vmax % c.v) bad("unsupported sampling factors");
            c.bw = mcux * c.h;
            c.bh = mcuy * c.v;
            c.cw = (W * c.h + hmax - 1) / hmax;
            c.ch = (H * c.v + vmax - 1) / vmax;
            c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
        progressive = marker == 0xC2;
        have_sof = true;
    }

    // ---- one scan ---------------------------------------------------------------------------------------------------------
    int eobrun = 0;
    void block_baseline(BitReader& br, Component& c, int16_t* b) {
        const Huff &hd = dc[c.td], &ha = ac[c.ta];
        const int s = decode(br, hd);
        if (s > 11) bad("bad DC difference");
        c.pred += s ? extend((int)br.get(s), s) : 0;
        b[0] = (int16_t)c.pred;
        for (int k = 1; k < 64;) {
            const int rs = decode(br, ha), r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (r != 15) break;
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) bad("coefficient index out of range");
            b[ZIGZAG[k]] = (int16_t)extend((int)br.get(sz), sz);
            ++k;
        }
    }
    void block_dc_first(BitReader& br, Component& c, int16_t* b, int al) {
        const int s = decode(br, dc[c.td]);
        if (s > 11) bad("bad DC difference");
        c.pred += s ? extend((int)br.get(s), s) : 0;
        b[0] = (int16_t)(c.pred * (1 << al));
    }
    void block_dc_refine(BitReader& br, int16_t* b, int al) {
        if (br.get(1)) b[0] = (int16_t)(b[0] | (1 << al));
    }
    void block_ac_first(BitReader& br, Component& c, int16_t* b, int ss, int se, int al) {
        if (eobrun > 0) { --eobrun; return; }
        const Huff& ha = ac[c.ta];
        for (int k = ss; k <= se;) {
            const int rs = decode(br, ha), r = rs >> 4, s = rs & 15;
            if (s) {
                k += r;
                if (k > 63) bad("coefficient index out of range");
                b[ZIGZAG[k]] = (int16_t)(extend((int)br.get(s), s) * (1 << al));
                ++k;
            } else if (r == 15) {
                k += 16;
            } else {
                eobrun = 1 << r;
                if (r) eobrun += (int)br.get(r);
                --eobrun;
                break;
            }
        }
    }
    void block_ac_refine(BitReader& br, Component& c, int16_t* b, int ss, int se, int al) {
        const int p1 = 1 << al, m1 = -(1 << al);
        const Huff& ha = ac[c.ta];
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se; ++k) {
                const int rs = decode(br, ha);
                int r = rs >> 4, s = rs & 15;
                if (s) {
                    if (s != 1) bad("bad refinement scan");
                    s = br.get(1) ? p1 : m1;
                } else if (r != 15) {
                    eobrun = 1 << r;
                    if (r) eobrun += (int)br.get(r);
                    break;                                            // the rest of the band is handled below
                }
                do {                                                  // skip r zero-history coefficients, correcting the others on the way
                    int16_t& co = b[ZIGZAG[k]];
                    if (co != 0) {
                        if (br.get(1) && (co & p1) == 0) co = (int16_t)(co + (co >= 0 ? p1 : m1));
                    } else if (--r < 0) {
                        break;
                    }
                    ++k;
                } while (k <= se);
                if (s) {
                    if (k > 63) bad("coefficient index out of range");
                    b[ZIGZAG[k]] = (int16_t)s;
                }
            }
        }
        if (eobrun > 0) {
            for (; k <= se; ++k) {
                int16_t& co = b[ZIGZAG[k]];
                if (co != 0 && br.get(1) && (co & p1) == 0) co = (int16_t)(co + (co >= 0 ? p1 : m1));
            }
            --eobrun;
        }
    }

    void read_scan(size_t p, size_t end) {
        if (!have_sof) bad("scan before the frame header");
        if (end - p < 1) bad("truncated scan header");
        const int ns = d[p];
        if (ns < 1 || ns > ncomp || end - p < 4 + 2 * (size_t)ns) bad("bad scan header");
        Component* sc[3];
        for (int i = 0; i < ns; ++i) {
            const int id = d[p + 1 + 2 * i];
            Component* c = nullptr;
            for (int j = 0; j < ncomp; ++j) if (comp[j].id == id) c = &comp[j];
            if (!c) bad("scan names an unknown component");
            for (int j = 0; j < i; ++j) if (sc[j] == c) bad("scan names a component twice");
            c->td = d[p + 2 + 2 * i] >> 4;
            c->ta = d[p + 2 + 2 * i] & 15;
            if (c->td > 3 || c->ta > 3) bad("bad table selector");
            sc[i] = c;
        }
        const int ss = d[p + 1 + 2 * ns], se = d[p + 2 + 2 * ns], ah = d[p + 3 + 2 * ns] >> 4, al = d[p + 3 + 2 * ns] & 15;
        if (progressive) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13 || ah > 13) bad("bad progressive scan parameters");
        } else if (ss != 0 || se != 63 || ah != 0 || al != 0) {
            bad("bad sequential scan parameters");
        }
        for (int i = 0; i < ns; ++i) {
            if ((!progressive || ss == 0) && !(progressive && ah) && !dc[sc[i]->td].present) bad("scan uses a missing DC table");
            if ((!progressive || ss > 0) && !ac[sc[i]->ta].present) bad("scan uses a missing AC table");
            if (!have_qt[sc[i]->tq]) bad("component uses a missing quantisation table");
        }
        BitReader br{d + end, d + n};
        for (int j = 0; j < ncomp; ++j) comp[j].pred = 0;
        eobrun = 0;
        auto one = [&](Component& c, int16_t* b) {
            if (!progressive) block_baseline(br, c, b);
            else if (ss == 0) { if (ah == 0) block_dc_first(br, c, b, al); else block_dc_refine(br, b, al); }
            else if (ah == 0) block_ac_first(br, c, b, ss, se, al);
            else block_ac_refine(br, c, b, ss, se, al);
        };
        int todo = restart_interval;
        auto maybe_restart = [&](bool last) {
            if (!restart_interval || --todo > 0 || last) return;
            // byte-align, expect RSTn; libjpeg tolerates damage here, this decoder only resynchronises on a marker it has seen
            if (!br.marker) {
                br.n = 0; br.acc = 0;
                br.fill();
            }
            if (br.marker < 0xD0 || br.marker > 0xD7) bad("missing restart marker");
            br.restart();
            for (int j = 0; j < ncomp; ++j) comp[j].pred = 0;
            eobrun = 0;
            todo = restart_interval;
        };
        if (ns == 1) {   // non-interleaved: the component's own block grid (real size, not the MCU padding)
            Component& c = *sc[0];
            const int nbx = (c.cw + 7) / 8, nby = (c.ch + 7) / 8;
            for (int by = 0; by < nby; ++by)
                for (int bx = 0; bx < nbx; ++bx) {
                    one(c, &c.coef[((size_t)by * c.bw + bx) * 64]);
                    maybe_restart(by == nby - 1 && bx == nbx - 1);
                }
        } else {
            for (int my = 0; my < mcuy; ++my)
                for (int mx = 0; mx < mcux; ++mx) {
                    for (int i = 0; i < ns; ++i) {
                        Component& c = *sc[i];
                        for (int v = 0; v < c.v; ++v)
                            for (int h = 0; h < c.h; ++h) one(c, &c.coef[((size_t)(my * c.v + v) * c.bw + (mx * c.h + h)) * 64]);
                    }
                    maybe_restart(my == mcuy - 1 && mx == mcux - 1);
                }
        }
        // continue behind the entropy-coded segment: at the marker the bit reader ran into, or the next one in the byte stream
        size_t q = (size_t)(br.p - d);
        if (br.marker) {
            pos = q - 2;
        } else {
            while (q + 1 < n && !(d[q] == 0xFF && d[q + 1] != 0 && (d[q + 1] < 0xD0 || d[q + 1] > 0xD7))) ++q;
            pos = q;
        }
    }

    void parse() {
        if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) bad("not a JPEG file");
        bool eoi = false;
        int scans = 0;
        while (!eoi) {
            if (pos + 2 > n) break;                                   // no EOI: decode what the scans gave (libjpeg does the same, with a warning)
            if (d[pos] != 0xFF) { ++pos; continue; }
            const int m = d[pos + 1];
            if (m == 0xFF) { ++pos; continue; }
            pos += 2;
            if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if (m == 0xD9) { eoi = true; break; }
            const size_t len = be16(pos);
            if (len < 2 || pos + len > n) bad("truncated segment");
            const size_t p = pos + 2, end = pos + len;
            switch (m) {
            case 0xDB: read_dqt(p, end); break;
            case 0xC4: read_dht(p, end); break;
            case 0xC0: case 0xC1: case 0xC2: read_sof(p, end, m); break;
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                bad("unsupported JPEG process (lossless, hierarchical or arithmetic coding)");
            case 0xDD:
                if (len < 4) bad("bad restart interval");
                restart_interval = (int)be16(p);
                break;
            case 0xEE:
                if (len >= 14 && memcmp(d + p, "Adobe", 5) == 0) adobe_transform = d[p + 11];
                break;
            case 0xDA:
                if (++scans > 1000) bad("too many scans");
                pos = end;
                read_scan(p, end);
                continue;
            default: break;                                           // APPn, COM, ...
            }
            pos = end;
        }
        if (!have_sof || scans == 0) bad("no image data");
    }

    // ---- coefficients -> samples -> RGB -------------------------------------------------------------------------------------
    std::vector<uint8_t> plane(const Component& c) const {   // bw*8 x bh*8 samples
        std::vector<uint8_t> out((size_t)c.bw * 8 * c.bh * 8);
        const int stride = c.bw * 8;
        for (int by = 0; by < c.bh; ++by)
            for (int bx = 0; bx < c.bw; ++bx) idct_islow(&c.coef[((size_t)by * c.bw + bx) * 64], qt[c.tq], &out[(size_t)by * 8 * stride + (size_t)bx * 8], stride);
        return out;
    }
    // jdsample.c: "fancy" (triangle filter) upsampling for 2:1 ratios, sample replication for 1:1. The component's REAL size is
    // cw x ch; rows / columns beyond the image edge are the edge's replicas (jdmainct.c), not the padding blocks' content.
    std::vector<uint8_t> upsample(const Component& c, const std::vector<uint8_t>& in) const {
        const int stride = c.bw * 8, fx = hmax / c.h, fy = vmax / c.v;
        std::vector<uint8_t> out((size_t)W * H);
        if (fx == 1 && fy == 1) {
            for (int y = 0; y < H; ++y) memcpy(&out[(size_t)y * W], &in[(size_t)y * stride], (size_t)W);
            return out;
        }
        const int cw = c.cw, ch = c.ch;
        auto row = [&](int y) { return &in[(size_t)std::min(std::max(y, 0), ch - 1) * stride]; };
        std::vector<int> sum((size_t)cw);
        for (int y = 0; y < H; ++y) {
            uint8_t* o = &out[(size_t)y * W];
            if (fy == 1) {                       // h2v1
                const uint8_t* r = row(y);
                for (int x = 0; x < W; ++x) {
                    const int i = x >> 1;
                    if (cw == 1) { o[x] = r[0]; continue; }
                    if ((x & 1) == 0) o[x] = i == 0 ? r[0] : (uint8_t)((r[i] * 3 + r[i - 1] + 1) >> 2);
                    else o[x] = i == cw - 1 ? r[i] : (uint8_t)((r[i] * 3 + r[i + 1] + 2) >> 2);
                }
            } else {                             // h1v2 / h2v2: the nearer input row weighs 3, the farther 1
                const int iy = y >> 1;
                const uint8_t *r0 = row(iy), *r1 = row((y & 1) ? iy + 1 : iy - 1);
                if (fx == 1) {                   // h1v2 (libjpeg-turbo: (3 near + far + 1 or 2) >> 2, bias 1 for the upper output row, 2 for the lower)
                    const int bias = (y & 1) ? 2 : 1;
                    for (int x = 0; x < W; ++x) o[x] = (uint8_t)((r0[x] * 3 + r1[x] + bias) >> 2);
                    continue;
                }
                for (int i = 0; i < cw; ++i) sum[(size_t)i] = r0[i] * 3 + r1[i];
                for (int x = 0; x < W; ++x) {
                    const int i = x >> 1, s = sum[(size_t)i];
                    if (cw == 1) { o[x] = (uint8_t)((s * 4 + 8) >> 4); continue; }
                    if ((x & 1) == 0) o[x] = i == 0 ? (uint8_t)((s * 4 + 8) >> 4) : (uint8_t)((s * 3 + sum[(size_t)i - 1] + 8) >> 4);
                    else o[x] = i == cw - 1 ? (uint8_t)((s * 4 + 7) >> 4) : (uint8_t)((s * 3 + sum[(size_t)i + 1] + 7) >> 4);
                }
            }
        }
        return out;
    }

    std::vector<uint8_t> to_rgb() const {
        std::vector<uint8_t> rgb((size_t)W * H * 3);
        std::vector<uint8_t> pl[3];
        for (int i = 0; i < ncomp; ++i) pl[i] = upsample(comp[i], plane(comp[i]));
        if (ncomp == 1) {
            for (size_t i = 0; i < (size_t)W * H; ++i) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = pl[0][i];
            return rgb;
        }
        const bool is_rgb = adobe_transform == 0 || (adobe_transform < 0 && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B');
        if (is_rgb) {
            for (size_t i = 0; i < (size_t)W * H; ++i) { rgb[3 * i] = pl[0][i]; rgb[3 * i + 1] = pl[1][i]; rgb[3 * i + 2] = pl[2][i]; }
            return rgb;
        }
        // jdcolor.c build_ycc_rgb_table: 16-bit fixed point, the green contributions kept unrounded until they are summed
        int cr_r[256], cb_b[256];
        long cr_g[256], cb_g[256];
        for (int i = 0; i < 256; ++i) {
            const long x = i - 128;
            cr_r[i] = (int)((91881L * x + 32768L) >> 16);      // FIX(1.40200)
            cb_b[i] = (int)((116130L * x + 32768L) >> 16);     // FIX(1.77200)
            cr_g[i] = -46802L * x;                             // FIX(0.71414)
            cb_g[i] = -22554L * x + 32768L;                    // FIX(0.34414), rounding term included
        }
        auto clamp = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        for (size_t i = 0; i < (size_t)W * H; ++i) {
            const int y = pl[0][i], cb = pl[1][i], cr = pl[2][i];
            rgb[3 * i] = clamp(y + cr_r[cr]);
            rgb[3 * i + 1] = clamp(y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
            rgb[3 * i + 2] = clamp(y + cb_b[cb]);
        }
        return rgb;
    }
};

}  // namespace

extern "C" int pt_load_jpeg_rgb8(const char* path, uint8_t** rgb, uint32_t* w, uint32_t* h) {
    try {
        std::ifstream in(path, std::ios::binary);
        if (!in) return set_error(std::string("pt_load_jpeg_rgb8: cannot open ") + path);
        std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        Decoder dec{f.data(), f.size()};
        dec.parse();
        std::vector<uint8_t> out = dec.to_rgb();
        uint8_t* p = (uint8_t*)malloc(out.size() + 4);
        if (!p) return set_error("pt_load_jpeg_rgb8: out of memory");
        memcpy(p, out.data(), out.size());
        *rgb = p;
        *w = (uint32_t)dec.W;
        *h = (uint32_t)dec.H;
        return 0;
    } catch (const std::exception& e) {
        return set_error(std::string("pt_load_jpeg_rgb8: ") + e.what());
    } catch (...) {
        return set_error("pt_load_jpeg_rgb8: unexpected failure");
    }
}
