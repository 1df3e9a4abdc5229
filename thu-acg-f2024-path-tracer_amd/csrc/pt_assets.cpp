// Asset ingest / output on the host: Wavefront OBJ (role of tobj::load_obj, main.rs:408),
// Radiance RGBE .hdr squashed to RGB8 (role of image's decode().to_rgb8(), texture.rs:62-67)
// and PNG output (imgbuf.save, camera.rs:118). None of this is on the hot path.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
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

// Positions and texcoords are parsed as f32 (tobj does); faces are fan-triangulated; only the
// position index of each `v/vt/vn` corner is kept, 1-based or negative-relative -> 0-based u32.
extern "C" int pt_load_obj(const char* path, float** pos, uint32_t* n_pos, uint32_t** idx, uint32_t* n_idx, float** uv,
                           uint32_t* n_uv) {
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

// RGBE -> f32 (mantissa * 2^(e-136), e == 0 -> 0) -> round(clamp(x,0,1)*255), the image
// crate's HDR -> Rgb8 conversion. Handles new-style per-channel RLE and flat scanlines.
extern "C" int pt_load_hdr_rgb8(const char* path, uint8_t** rgb, uint32_t* w, uint32_t* h) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return set_error(std::string("pt_load_hdr_rgb8: cannot open ") + path);
    std::string line;
    if (!std::getline(in, line) || line.compare(0, 2, "#?") != 0) return set_error("pt_load_hdr_rgb8: not a Radiance file");
    while (std::getline(in, line) && !line.empty()) {}
    if (!std::getline(in, line)) return set_error("pt_load_hdr_rgb8: missing resolution line");
    int W = 0, H = 0;
    if (sscanf(line.c_str(), "-Y %d +X %d", &H, &W) != 2 || W <= 0 || H <= 0) return set_error("pt_load_hdr_rgb8: unsupported orientation");
    std::vector<uint8_t> rest((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t p = 0;
    auto need = [&](size_t n) { return p + n <= rest.size(); };
    std::vector<uint8_t> out((size_t)W * H * 3), scan((size_t)W * 4);
    for (int y = 0; y < H; ++y) {
        if (!need(4)) return set_error("pt_load_hdr_rgb8: truncated file");
        const bool rle = rest[p] == 2 && rest[p + 1] == 2 && (rest[p + 2] & 0x80) == 0 && ((rest[p + 2] << 8) | rest[p + 3]) == W;
        if (rle) {
            p += 4;
            for (int c = 0; c < 4; ++c)
                for (int x = 0; x < W;) {
                    if (!need(1)) return set_error("pt_load_hdr_rgb8: truncated file");
                    int n = rest[p++];
                    if (n > 128) {
                        n -= 128;
                        if (!need(1) || x + n > W) return set_error("pt_load_hdr_rgb8: bad run");
                        uint8_t v = rest[p++];
                        while (n--) scan[(size_t)(x++) * 4 + c] = v;
                    } else {
                        if (!need((size_t)n) || x + n > W) return set_error("pt_load_hdr_rgb8: bad run");
                        while (n--) scan[(size_t)(x++) * 4 + c] = rest[p++];
                    }
                }
        } else {
            if (!need((size_t)W * 4)) return set_error("pt_load_hdr_rgb8: truncated file");
            memcpy(scan.data(), &rest[p], (size_t)W * 4);
            p += (size_t)W * 4;
        }
        for (int x = 0; x < W; ++x) {
            const uint8_t* q = &scan[(size_t)x * 4];
            for (int c = 0; c < 3; ++c) {
                float v = q[3] ? (float)q[c] * std::ldexp(1.0f, (int)q[3] - 136) : 0.0f;
                v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                out[((size_t)y * W + x) * 3 + c] = (uint8_t)std::round(v * 255.0f);
            }
        }
    }
    *rgb = to_malloc(out);
    *w = (uint32_t)W;
    *h = (uint32_t)H;
    return 0;
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
