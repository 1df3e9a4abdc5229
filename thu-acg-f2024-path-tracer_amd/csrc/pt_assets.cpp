// Asset ingest / output on the host: Wavefront OBJ (role of tobj::load_obj, main.rs:408),
// Radiance RGBE .hdr squashed to RGB8 (role of image's decode().to_rgb8(), texture.rs:62-67)
// and PNG output (imgbuf.save, camera.rs:118). None of this is on the hot path.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pt_amd.h"
#include "pt_scene.h"

using pt::set_error;

template <class T>
static T* to_malloc(const std::vector<T>& v) {
    T* p = (T*)malloc(v.size() * sizeof(T) + sizeof(T));
    if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

// Untrusted files: decoded sizes are bounded before anything is allocated, and no C++ exception (std::bad_alloc from a hostile
// header, std::length_error) crosses the C ABI — every extern "C" loader of this file runs its body through guarded().
static const uint64_t MAX_IMAGE_PIXELS = 1ull << 28;   // 268 M pixels: 3 GiB as f32 RGB — far above any asset, far below a size_t overflow
// every extern "C" loader runs inside this: a std::bad_alloc / length_error from a hostile header must not cross the C ABI
template <class F>
static int guarded(const char* who, F&& f) {
    try {
        return f();
    } catch (const std::exception& e) {
        return set_error(std::string(who) + ": " + e.what());
    } catch (...) {
        return set_error(std::string(who) + ": unexpected failure");
    }
}
// Positions and texcoords are parsed as f32 (tobj does); faces are fan-triangulated; only the
// position index of each `v/vt/vn` corner is kept, 1-based or negative-relative -> 0-based u32.
static int load_obj(const char* path, float** pos, uint32_t* n_pos, uint32_t** idx, uint32_t* n_idx, float** uv, uint32_t* n_uv);
extern "C" int pt_load_obj(const char* path, float** pos, uint32_t* n_pos, uint32_t** idx, uint32_t* n_idx, float** uv,
                           uint32_t* n_uv) {
    return guarded("pt_load_obj", [&]() { return load_obj(path, pos, n_pos, idx, n_idx, uv, n_uv); });
}
static int load_obj(const char* path, float** pos, uint32_t* n_pos, uint32_t** idx, uint32_t* n_idx, float** uv, uint32_t* n_uv) {
    std::ifstream in(path);
    if (!in) return set_error(std::string("pt_load_obj: cannot open ") + path);
    std::vector<float> P, T;
    std::vector<uint32_t> I;
    std::string line, tag, corner;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        if (!(ls >> tag)) continue;
        if (tag == "v") {
            std::string a, b, c;
            if (!(ls >> a >> b >> c)) return set_error("pt_load_obj: malformed vertex");
            P.push_back(strtof(a.c_str(), nullptr));
            P.push_back(strtof(b.c_str(), nullptr));
            P.push_back(strtof(c.c_str(), nullptr));
        } else if (tag == "vt") {
            std::string a, b;
            if (!(ls >> a >> b)) return set_error("pt_load_obj: malformed texcoord");
            T.push_back(strtof(a.c_str(), nullptr));
            T.push_back(strtof(b.c_str(), nullptr));
        } else if (tag == "f") {
            std::vector<uint32_t> poly;
            const long nv = (long)(P.size() / 3);
            while (ls >> corner) {
                long v = strtol(corner.c_str(), nullptr, 10);   // stops at '/'
                v = v < 0 ? nv + v : v - 1;
                if (v < 0 || v >= nv) return set_error("pt_load_obj: face index out of range");
                poly.push_back((uint32_t)v);
            }
            for (size_t k = 1; k + 1 < poly.size(); ++k) {
                I.push_back(poly[0]);
                I.push_back(poly[k]);
                I.push_back(poly[k + 1]);
            }
        }
    }
    *pos = to_malloc(P); *n_pos = (uint32_t)(P.size() / 3);
    *idx = to_malloc(I); *n_idx = (uint32_t)I.size();
    *uv = to_malloc(T); *n_uv = (uint32_t)(T.size() / 2);
    return 0;
}

// OBJ with `vn` and separate v/vt/vn index streams: single-index expansion (what tobj does with `single_index: true`).
// The reference indexes normals and texcoords with the POSITION index (mesh.rs:173-184, under tobj's
// OFFLINE_RENDERING_LOAD_OPTIONS) — right only for files whose three index streams coincide. Here every distinct
// (v, vt, vn) corner becomes one output vertex, in order of first appearance, so that pt_mesh's position-indexed
// attribute lookup is right for any file. n_nrm / n_uv are 0 when the file has no vn / vt, or when a face omits them.
#include <map>
#include <tuple>
static int load_obj_single_index(const char* path, float** pos, uint32_t* n_pos, uint32_t** idx, uint32_t* n_idx, float** nrm, uint32_t* n_nrm,
                                 float** uv, uint32_t* n_uv);
extern "C" int pt_load_obj_single_index(const char* path, float** pos, uint32_t* n_pos, uint32_t** idx, uint32_t* n_idx, float** nrm,
                                        uint32_t* n_nrm, float** uv, uint32_t* n_uv) {
    return guarded("pt_load_obj_single_index", [&]() { return load_obj_single_index(path, pos, n_pos, idx, n_idx, nrm, n_nrm, uv, n_uv); });
}
static int load_obj_single_index(const char* path, float** pos, uint32_t* n_pos, uint32_t** idx, uint32_t* n_idx, float** nrm, uint32_t* n_nrm,
                                 float** uv, uint32_t* n_uv) {
    std::ifstream in(path);
    if (!in) return set_error(std::string("pt_load_obj_single_index: cannot open ") + path);
    std::vector<float> P, T, N, oP, oT, oN;
    std::vector<uint32_t> I;
    std::map<std::tuple<long, long, long>, uint32_t> seen;
    bool all_uv = true, all_nrm = true;
    std::string line, tag, corner;
    auto f3 = [&](std::istringstream& ls, std::vector<float>& dst, int n) {
        for (int k = 0; k < n; ++k) {
            std::string a;
            if (!(ls >> a)) return false;
            dst.push_back(strtof(a.c_str(), nullptr));
        }
        return true;
    };
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        if (!(ls >> tag)) continue;
        if (tag == "v") { if (!f3(ls, P, 3)) return set_error("pt_load_obj_single_index: malformed vertex"); }
        else if (tag == "vt") { if (!f3(ls, T, 2)) return set_error("pt_load_obj_single_index: malformed texcoord"); }
        else if (tag == "vn") { if (!f3(ls, N, 3)) return set_error("pt_load_obj_single_index: malformed normal"); }
        else if (tag == "f") {
            std::vector<uint32_t> poly;
            const long nv = (long)(P.size() / 3), nt = (long)(T.size() / 2), nn = (long)(N.size() / 3);
            while (ls >> corner) {
                long c[3] = {0, 0, 0};   // 1-based v, vt, vn; 0 = absent
                const char* q = corner.c_str();
                for (int k = 0; k < 3 && *q; ++k) {
                    char* end;
                    c[k] = strtol(q, &end, 10);
                    q = end;
                    if (*q == '/') ++q;
                }
                auto fix = [](long i, long n) { return i < 0 ? n + i : i - 1; };   // negative = relative to the end
                const long v = fix(c[0], nv), t = c[1] ? fix(c[1], nt) : -1, n = c[2] ? fix(c[2], nn) : -1;
                if (v < 0 || v >= nv || t >= nt || n >= nn || (c[1] && t < 0) || (c[2] && n < 0)) return set_error("pt_load_obj_single_index: face index out of range");
                all_uv = all_uv && t >= 0;
                all_nrm = all_nrm && n >= 0;
                auto key = std::make_tuple(v, t, n);
                auto it = seen.find(key);
                if (it == seen.end()) {
                    it = seen.emplace(key, (uint32_t)(oP.size() / 3)).first;
                    for (int k = 0; k < 3; ++k) oP.push_back(P[3 * v + k]);
                    for (int k = 0; k < 2; ++k) oT.push_back(t >= 0 ? T[2 * t + k] : 0.0f);
                    for (int k = 0; k < 3; ++k) oN.push_back(n >= 0 ? N[3 * n + k] : 0.0f);
                }
                poly.push_back(it->second);
            }
            for (size_t k = 1; k + 1 < poly.size(); ++k) { I.push_back(poly[0]); I.push_back(poly[k]); I.push_back(poly[k + 1]); }
        }
    }
    if (!all_uv || T.empty()) oT.clear();
    if (!all_nrm || N.empty()) oN.clear();
    *pos = to_malloc(oP); *n_pos = (uint32_t)(oP.size() / 3);
    *idx = to_malloc(I); *n_idx = (uint32_t)I.size();
    *nrm = to_malloc(oN); *n_nrm = (uint32_t)(oN.size() / 3);
    *uv = to_malloc(oT); *n_uv = (uint32_t)(oT.size() / 2);
    return 0;
}

// PNG -> RGB8 (role of image's PngDecoder + to_rgb8(), texture.rs:62-67: alpha dropped, grey replicated, palettes expanded,
// 16-bit samples reduced as image 0.25.5 converts u16 -> u8: (v + 128) / 257, i.e. rounded, not truncated). zlib inflates the IDAT stream; the five scanline filters are undone here.
// Interlaced (Adam7) files are rejected.
static int load_png_rgb8(const char* path, uint8_t** rgb, uint32_t* w, uint32_t* h) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return set_error(std::string("pt_load_png_rgb8: cannot open ") + path);
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (f.size() < 8 || memcmp(f.data(), sig, 8) != 0) return set_error("pt_load_png_rgb8: not a PNG file");
    auto be32 = [&](size_t p) { return ((uint32_t)f[p] << 24) | ((uint32_t)f[p + 1] << 16) | ((uint32_t)f[p + 2] << 8) | f[p + 3]; };
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    for (size_t p = 8; p + 12 <= f.size();) {
        const uint32_t len = be32(p);
        if (p + 12 + (size_t)len > f.size()) return set_error("pt_load_png_rgb8: truncated chunk");
        const std::string type((const char*)&f[p + 4], 4);
        const uint8_t* d = &f[p + 8];
        if (crc32(crc32(0, nullptr, 0), &f[p + 4], len + 4) != be32(p + 8 + len)) return set_error("pt_load_png_rgb8: chunk CRC mismatch");
        if (type == "IHDR") {
            if (len < 13) return set_error("pt_load_png_rgb8: bad IHDR");
            W = be32(p + 8); H = be32(p + 12);
            depth = d[8]; ctype = d[9]; interlace = d[12];
        } else if (type == "PLTE") plte.assign(d, d + len);
        else if (type == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (type == "IEND") break;
        p += 12 + (size_t)len;
    }
    if (W == 0 || H == 0 || (uint64_t)W * H > MAX_IMAGE_PIXELS) return set_error("pt_load_png_rgb8: bad image size");
    if (idat.size() > 0xFFFFFFFFull) return set_error("pt_load_png_rgb8: image data too large");
    if (interlace) return set_error("pt_load_png_rgb8: interlaced PNG is not supported");
    const int chans = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!chans || !(depth == 8 || depth == 16 || (depth < 8 && (ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))))
        return set_error("pt_load_png_rgb8: unsupported colour type / bit depth");
    if (ctype == 3 && depth == 16) return set_error("pt_load_png_rgb8: a palette image cannot have 16-bit indices");
    if (ctype == 3 && plte.size() < 3) return set_error("pt_load_png_rgb8: palette image without PLTE");
    const size_t bpp = std::max<size_t>(1, (size_t)chans * depth / 8), stride = ((size_t)W * chans * depth + 7) / 8;
    std::vector<uint8_t> raw((stride + 1) * H);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) return set_error("pt_load_png_rgb8: zlib stream is damaged");
    std::vector<uint8_t> prev(stride, 0), out((size_t)W * H * 3);
    for (uint32_t y = 0; y < H; ++y) {
        uint8_t* row = &raw[(size_t)y * (stride + 1)];
        const int ft = row[0];
        uint8_t* cur = row + 1;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } else if (ft != 0) return set_error("pt_load_png_rgb8: bad filter type");
            cur[i] = (uint8_t)(cur[i] + pred);
        }
        memcpy(prev.data(), cur, stride);
        auto sample = [&](uint32_t x, int ch) -> uint8_t {   // 8-bit value of channel ch of pixel x
            if (depth == 8) return cur[(size_t)x * chans + ch];
            if (depth == 16) {   // big-endian u16 -> u8 like image's to_rgb8: (v + 128) / 257
                const size_t k = ((size_t)x * chans + ch) * 2;
                const uint32_t v16 = ((uint32_t)cur[k] << 8) | cur[k + 1];
                return (uint8_t)((v16 + 128u) / 257u);
            }
            const size_t bit = (size_t)x * depth;
            const int v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
            return ctype == 3 ? (uint8_t)v : (uint8_t)(v * 255 / ((1 << depth) - 1));
        };
        for (uint32_t x = 0; x < W; ++x) {
            uint8_t* o = &out[((size_t)y * W + x) * 3];
            if (ctype == 3) {
                const size_t k = sample(x, 0);
                if (3 * k + 2 >= plte.size()) return set_error("pt_load_png_rgb8: palette index out of range");
                o[0] = plte[3 * k]; o[1] = plte[3 * k + 1]; o[2] = plte[3 * k + 2];
            } else if (chans <= 2) {
                o[0] = o[1] = o[2] = sample(x, 0);
            } else {
                o[0] = sample(x, 0); o[1] = sample(x, 1); o[2] = sample(x, 2);
            }
        }
    }
    *rgb = to_malloc(out);
    *w = W; *h = H;
    return 0;
}
extern "C" int pt_load_png_rgb8(const char* path, uint8_t** rgb, uint32_t* w, uint32_t* h) {
    return guarded("pt_load_png_rgb8", [&]() { return load_png_rgb8(path, rgb, w, h); });
}

// Radiance RGBE -> f32 (mantissa * 2^(e-136), e == 0 -> 0), the image crate's HdrDecoder. Handles new-style per-channel RLE and
// flat scanlines. Untrusted input: the header's size is bounded before anything is allocated, every run is checked against the
// scanline and the file's end, and no exception leaves this file's extern "C" functions (guarded()).
static int load_hdr_f32(const char* who, const char* path, std::vector<float>& out, uint32_t* w, uint32_t* h) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return set_error(std::string(who) + ": cannot open " + path);
    std::string line;
    if (!std::getline(in, line) || line.compare(0, 2, "#?") != 0) return set_error(std::string(who) + ": not a Radiance file");
    while (std::getline(in, line) && !line.empty()) {}
    if (!std::getline(in, line)) return set_error(std::string(who) + ": missing resolution line");
    int W = 0, H = 0;
    if (sscanf(line.c_str(), "-Y %d +X %d", &H, &W) != 2 || W <= 0 || H <= 0) return set_error(std::string(who) + ": unsupported orientation");
    if ((uint64_t)W * (uint64_t)H > MAX_IMAGE_PIXELS) return set_error(std::string(who) + ": image too large");
    std::vector<uint8_t> rest((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t p = 0;
    auto need = [&](size_t n) { return n <= rest.size() && p <= rest.size() - n; };
    std::vector<uint8_t> scan((size_t)W * 4);
    out.assign((size_t)W * H * 3, 0.0f);
    for (int y = 0; y < H; ++y) {
        if (!need(4)) return set_error(std::string(who) + ": truncated file");
        const bool rle = rest[p] == 2 && rest[p + 1] == 2 && (rest[p + 2] & 0x80) == 0 && ((rest[p + 2] << 8) | rest[p + 3]) == W;
        if (rle) {
            p += 4;
            for (int c = 0; c < 4; ++c)
                for (int x = 0; x < W;) {
                    if (!need(1)) return set_error(std::string(who) + ": truncated file");
                    int n = rest[p++];
                    if (n > 128) {
                        n -= 128;
                        if (!need(1) || x + n > W) return set_error(std::string(who) + ": bad run");
                        uint8_t v = rest[p++];
                        while (n--) scan[(size_t)(x++) * 4 + c] = v;
                    } else {
                        if (n == 0 || !need((size_t)n) || x + n > W) return set_error(std::string(who) + ": bad run");
                        while (n--) scan[(size_t)(x++) * 4 + c] = rest[p++];
                    }
                }
        } else {
            if (!need((size_t)W * 4)) return set_error(std::string(who) + ": truncated file");
            memcpy(scan.data(), &rest[p], (size_t)W * 4);
            p += (size_t)W * 4;
        }
        for (int x = 0; x < W; ++x) {
            const uint8_t* q = &scan[(size_t)x * 4];
            for (int c = 0; c < 3; ++c) out[((size_t)y * W + x) * 3 + c] = q[3] ? (float)q[c] * std::ldexp(1.0f, (int)q[3] - 136) : 0.0f;
        }
    }
    *w = (uint32_t)W;
    *h = (uint32_t)H;
    return 0;
}
// ImageReader::open(..).decode() of a Radiance file: Rgb32F, kept as it is (the float-HDR option, pt_tex_image_rgbf32)
extern "C" int pt_load_hdr_rgbf32(const char* path, float** rgb, uint32_t* w, uint32_t* h) {
    return guarded("pt_load_hdr_rgbf32", [&]() {
        std::vector<float> v;
        if (load_hdr_f32("pt_load_hdr_rgbf32", path, v, w, h) != 0) return -1;
        *rgb = to_malloc(v);
        return 0;
    });
}
// ... followed by .to_rgb8() (texture.rs:62-67): round(clamp(x,0,1)*255)
extern "C" int pt_load_hdr_rgb8(const char* path, uint8_t** rgb, uint32_t* w, uint32_t* h) {
    return guarded("pt_load_hdr_rgb8", [&]() {
        std::vector<float> v;
        if (load_hdr_f32("pt_load_hdr_rgb8", path, v, w, h) != 0) return -1;
        std::vector<uint8_t> out(v.size());
        for (size_t i = 0; i < v.size(); ++i) {
            const float c = v[i] < 0.0f ? 0.0f : (v[i] > 1.0f ? 1.0f : v[i]);
            out[i] = (uint8_t)std::round(c * 255.0f);
        }
        *rgb = to_malloc(out);
        return 0;
    });
}
extern "C" void pt_free(void* p) { free(p); }

// Minimal PNG encoder: 8-bit RGB, filter 0, one zlib stream.
static void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
static void put_chunk(std::vector<uint8_t>& png, const char* type, const std::vector<uint8_t>& data) {
    put_be32(png, (uint32_t)data.size());
    size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    put_be32(png, (uint32_t)crc32(0L, &png[start], (uInt)(png.size() - start)));
}
extern "C" int pt_save_png(const char* path, uint32_t w, uint32_t h, const uint8_t* rgb) {
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (w * 3 + 1));
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return set_error("pt_save_png: deflate failed");
    z.resize(zlen);
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w);
    put_be32(ihdr, h);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
    put_chunk(png, "IHDR", ihdr);
    put_chunk(png, "IDAT", z);
    put_chunk(png, "IEND", {});
    FILE* f = fopen(path, "wb");
    if (!f) return set_error(std::string("pt_save_png: cannot open ") + path);   // the reference only prints this error (camera.rs:120-122)
    size_t n = fwrite(png.data(), 1, png.size(), f);
    fclose(f);
    return n == png.size() ? 0 : set_error("pt_save_png: short write");
}
