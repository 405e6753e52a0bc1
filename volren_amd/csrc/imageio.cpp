// imageio.cpp -- see imageio.h
#include "imageio.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace vr {

namespace {
struct File {
    FILE* f;
    explicit File(FILE* f_) : f(f_) {}
    ~File() { if (f) fclose(f); }
};
}  // namespace

void load_hdr(const std::string& path, std::vector<float>& rgb, int& w, int& h) {
    File file(fopen(path.c_str(), "rb"));
    FILE* f = file.f;
    if (!f) throw std::runtime_error("Unable to read file: " + path);
    char line[512];
    bool have_fmt = false;
    if (!fgets(line, sizeof line, f) || strncmp(line, "#?", 2) != 0) throw std::runtime_error("Not a Radiance HDR file: " + path);
    for (;;) {
        if (!fgets(line, sizeof line, f)) throw std::runtime_error("Truncated HDR header: " + path);
        if (line[0] == '\n') break;
        if (strncmp(line, "FORMAT=32-bit_rle_rgbe", 22) == 0) have_fmt = true;
    }
    if (!fgets(line, sizeof line, f)) throw std::runtime_error("Truncated HDR header: " + path);
    int W = 0, H = 0;
    if (!have_fmt || sscanf(line, "-Y %d +X %d", &H, &W) != 2 || W <= 0 || H <= 0)
        throw std::runtime_error("Unsupported HDR layout (need FORMAT=32-bit_rle_rgbe, -Y h +X w): " + path);
    rgb.assign((size_t)W * H * 3, 0.f);
    std::vector<uint8_t> scan((size_t)W * 4);
    auto need = [&](void* dst, size_t n) { if (fread(dst, 1, n, f) != n) throw std::runtime_error("Truncated HDR data: " + path); };
    for (int y = 0; y < H; ++y) {
        uint8_t hd[4];
        need(hd, 4);
        if (W >= 8 && W < 32768 && hd[0] == 2 && hd[1] == 2 && !(hd[2] & 0x80)) {
            if ((((int)hd[2] << 8) | hd[3]) != W) throw std::runtime_error("Bad HDR scanline width: " + path);
            for (int ch = 0; ch < 4; ++ch) {
                int x = 0;
                while (x < W) {
                    uint8_t cnt; need(&cnt, 1);
                    if (cnt > 128) {
                        int n = cnt - 128; uint8_t val; need(&val, 1);
                        if (x + n > W) throw std::runtime_error("Bad HDR run: " + path);
                        while (n--) scan[4 * (size_t)(x++) + ch] = val;
                    } else {
                        int n = cnt;
                        if (n == 0 || x + n > W) throw std::runtime_error("Bad HDR run: " + path);
                        while (n--) { uint8_t val; need(&val, 1); scan[4 * (size_t)(x++) + ch] = val; }
                    }
                }
            }
        } else {
            memcpy(scan.data(), hd, 4);
            need(scan.data() + 4, (size_t)(W - 1) * 4);
        }
        for (int x = 0; x < W; ++x) {
            const uint8_t e = scan[4 * (size_t)x + 3];
            const float s = e ? std::ldexp(1.0f, (int)e - 136) : 0.0f;
            float* px = &rgb[3 * ((size_t)y * W + x)];
            px[0] = (float)scan[4 * (size_t)x + 0] * s;
            px[1] = (float)scan[4 * (size_t)x + 1] * s;
            px[2] = (float)scan[4 * (size_t)x + 2] * s;
        }
    }
    w = W; h = H;
}

static void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
static void chunk(std::vector<uint8_t>& png, const char* type, const uint8_t* data, size_t n) {
    put_be32(png, (uint32_t)n);
    const size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    if (n) png.insert(png.end(), data, data + n);
    put_be32(png, (uint32_t)crc32(0L, png.data() + start, (uInt)(n + 4)));
}

void save_png_rgba8(const std::string& path, const uint8_t* rgba, int w, int h) {
    std::vector<uint8_t> raw((size_t)h * ((size_t)w * 4 + 1));
    for (int y = 0; y < h; ++y) {
        raw[(size_t)y * ((size_t)w * 4 + 1)] = 0;      // filter: none
        memcpy(&raw[(size_t)y * ((size_t)w * 4 + 1) + 1], rgba + (size_t)y * w * 4, (size_t)w * 4);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw std::runtime_error("PNG: deflate failed");
    std::vector<uint8_t> png = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w); put_be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(png, "IHDR", ihdr.data(), ihdr.size());
    chunk(png, "IDAT", z.data(), zlen);
    chunk(png, "IEND", nullptr, 0);
    File file(fopen(path.c_str(), "wb"));
    if (!file.f || fwrite(png.data(), 1, png.size(), file.f) != png.size()) throw std::runtime_error("Unable to write file: " + path);
}

void framebuffer_to_rgba8(const float* fb, int w, int h, std::vector<uint8_t>& out) {
    out.resize((size_t)w * h * 4);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int k = 0; k < 4; ++k) {
                float v = fb[4 * ((size_t)(h - 1 - y) * w + x) + k];
                v = v != v ? 0.f : (v < 0.f ? 0.f : (v > 1.f ? 1.f : v));
                out[4 * ((size_t)y * w + x) + k] = (uint8_t)(v * 255.0f + 0.5f);
            }
}

}  // namespace vr
