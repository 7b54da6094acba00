// Output conveniences of the driver (main.rs:110-112, 121, 127-128): the PNG file the reference saves through
// `image::RgbImage::save`, and its time-stamped file name.  Plain 8-bit RGB, no interlace, one IDAT; the pixel
// bytes are what matters for parity, the compressed stream is zlib's.
#include "../../include/rtow_host.h"
#include "../../include/rtow_mi355x.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

namespace {
void put_be32(std::vector<unsigned char>& v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24)), v.push_back((unsigned char)(x >> 16)), v.push_back((unsigned char)(x >> 8)),
        v.push_back((unsigned char)x);
}
void chunk(std::vector<unsigned char>& out, const char type[4], const unsigned char* data, size_t n) {
    put_be32(out, (uint32_t)n);
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    put_be32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(n + 4)));
}
} // namespace

extern "C" {

int rth_png_write(const char* path, const uint8_t* rgb8, uint32_t nx, uint32_t ny) {
    if (!path || !rgb8 || nx == 0 || ny == 0 || (uint64_t)nx * ny > (1ull << 28)) return RT_ERR_INVALID;
    // filter type 0 in front of every scanline
    std::vector<unsigned char> raw((size_t)ny * (3 * (size_t)nx + 1));
    for (uint32_t j = 0; j < ny; ++j) {
        raw[(size_t)j * (3 * (size_t)nx + 1)] = 0;
        std::memcpy(&raw[(size_t)j * (3 * (size_t)nx + 1) + 1], rgb8 + (size_t)j * 3 * nx, 3 * (size_t)nx);
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) return RT_ERR_DEVICE;
    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, nx), put_be32(ihdr, ny);
    ihdr.push_back(8), ihdr.push_back(2), ihdr.push_back(0), ihdr.push_back(0), ihdr.push_back(0); // 8-bit, RGB
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    chunk(out, "IDAT", z.data(), zn);
    chunk(out, "IEND", nullptr, 0);
    // write to a temporary and rename, so a viewer following the preview never sees half a file
    const std::string tmp = std::string(path) + ".part";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return RT_ERR_INVALID;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), path) != 0) {
        std::remove(tmp.c_str());
        return RT_ERR_DEVICE;
    }
    return RT_OK;
}

int rth_output_file_name(int64_t unix_seconds, char* buf, uint32_t buf_len) {
    // main.rs:110-112: Local::now().to_rfc3339().replace(":", "-"), cut at the first '.', + ".png"
    //   2024-05-17T21:03:44.123456789+02:00 -> 2024-05-17T21-03-44.png
    if (!buf || buf_len < 24) return RT_ERR_INVALID;
    std::time_t t = unix_seconds < 0 ? std::time(nullptr) : (std::time_t)unix_seconds;
    std::tm tmv;
    if (!localtime_r(&t, &tmv)) return RT_ERR_INVALID;
    const int n = std::snprintf(buf, buf_len, "%04d-%02d-%02dT%02d-%02d-%02d.png", tmv.tm_year + 1900, tmv.tm_mon + 1, tmv.tm_mday,
                                tmv.tm_hour, tmv.tm_min, tmv.tm_sec);
    return (n > 0 && (uint32_t)n < buf_len) ? RT_OK : RT_ERR_INVALID;
}

} // extern "C"
