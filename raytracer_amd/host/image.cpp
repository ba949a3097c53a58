#include "image.h"

#include <cstdio>
#include <vector>

#include "log.h"

namespace luma {

Image::Image(std::int32_t width, std::int32_t height, bool allocate) noexcept
    : _width{ width }, _height{ height },
      _data{ allocate ? new std::uint32_t[static_cast<std::size_t>(width) * height]() : nullptr }, _owned{ allocate }
{
}

Image::~Image()
{
    if (_owned) { delete[] _data; }
}

void Image::shadow_from(std::uint32_t* data) noexcept
{
    if (_owned) { delete[] _data; _owned = false; }
    _data = data;     // lent, never freed here (the reference frees it after export, image.cpp:56)
}

namespace {

// Appends the decimal digits of v (0..255 in practice) to out.
inline char* put_u32(char* out, std::uint32_t v)
{
    char tmp[10];
    int n = 0;
    do { tmp[n++] = static_cast<char>('0' + v % 10); v /= 10; } while (v);
    while (n) { *out++ = tmp[--n]; }
    return out;
}

}  // namespace

// Same bytes as the reference's writer (image.cpp:30-64): "P3\nW H\n255\n", then one text line
// per x holding the H pixels of that column as "r g b  ".  The reference builds the whole file
// in one std::string (400 MB of text for an 8K frame); this streams it a column at a time.
void Image::export_to_ppm(const std::string& filepath) noexcept
{
    std::FILE* file = _data ? std::fopen(filepath.c_str(), "wb") : nullptr;
    if (!file) {
        log("error writing to file `" + filepath + "`");
        return;
    }
    std::fprintf(file, "P3\n%d %d\n255\n", _width, _height);
    std::vector<char> line(static_cast<std::size_t>(_height) * 14 + 1);   // "255 255 255  " = 13 chars
    for (std::int32_t x = 0; x < _width; x++) {
        char* out = line.data();
        for (std::int32_t y = 0; y < _height; y++) {
            const std::uint32_t px = _data[static_cast<std::size_t>(y) * _width + x];
            out = put_u32(out, (px >> 16) & 0xFFu); *out++ = ' ';
            out = put_u32(out, (px >> 8) & 0xFFu);  *out++ = ' ';
            out = put_u32(out, px & 0xFFu);         *out++ = ' '; *out++ = ' ';
        }
        *out++ = '\n';
        std::fwrite(line.data(), 1, static_cast<std::size_t>(out - line.data()), file);
    }
    std::fclose(file);
}

// Binary P6: 3 bytes per pixel, rows top to bottom (the standard raster order, unlike the
// column-major text of export_to_ppm).  An 8K frame is 99.5 MB instead of ~400 MB of text.
void Image::export_to_pnm(const std::string& filepath) noexcept
{
    std::FILE* file = _data ? std::fopen(filepath.c_str(), "wb") : nullptr;
    if (!file) {
        log("error writing to file `" + filepath + "`");
        return;
    }
    std::fprintf(file, "P6\n%d %d\n255\n", _width, _height);
    std::vector<unsigned char> row(static_cast<std::size_t>(_width) * 3u);
    for (std::int32_t y = 0; y < _height; ++y) {
        const std::uint32_t* src = _data + static_cast<std::size_t>(y) * _width;
        for (std::int32_t x = 0; x < _width; ++x) {
            row[3 * x + 0] = static_cast<unsigned char>((src[x] >> 16) & 0xFF);
            row[3 * x + 1] = static_cast<unsigned char>((src[x] >> 8) & 0xFF);
            row[3 * x + 2] = static_cast<unsigned char>(src[x] & 0xFF);
        }
        std::fwrite(row.data(), 1, row.size(), file);
    }
    std::fclose(file);
}

// 24-bit uncompressed BMP, rows bottom-up and padded to 4 bytes (image.cpp:66-153).
void Image::export_to_bmp(const std::string& filepath) noexcept
{
    std::FILE* file = _data ? std::fopen(filepath.c_str(), "wb") : nullptr;
    if (!file) {
        log("error writing to file `" + filepath + "`");
        return;
    }
    const std::uint32_t row_bytes = static_cast<std::uint32_t>(_width) * 3u;
    const std::uint32_t padded = (row_bytes + 3u) & ~3u;
    const std::uint32_t header = 14u + 40u;
    const std::uint32_t total = header + padded * static_cast<std::uint32_t>(_height);

    unsigned char h[54] = { 0 };
    auto le32 = [&h](int at, std::uint32_t v) { for (int i = 0; i < 4; i++) { h[at + i] = static_cast<unsigned char>(v >> (8 * i)); } };
    h[0] = 'B'; h[1] = 'M';
    le32(2, total);
    le32(10, header);
    le32(14, 40u);
    le32(18, static_cast<std::uint32_t>(_width));
    le32(22, static_cast<std::uint32_t>(_height));
    h[26] = 1;      // planes
    h[28] = 24;     // bits per pixel
    std::fwrite(h, 1, sizeof h, file);

    std::vector<unsigned char> row(padded, 0);
    for (std::int32_t y = _height - 1; y >= 0; --y) {
        const std::uint32_t* src = _data + static_cast<std::size_t>(y) * _width;
        for (std::int32_t x = 0; x < _width; ++x) {
            row[3 * x + 0] = static_cast<unsigned char>(src[x] & 0xFF);
            row[3 * x + 1] = static_cast<unsigned char>((src[x] >> 8) & 0xFF);
            row[3 * x + 2] = static_cast<unsigned char>((src[x] >> 16) & 0xFF);
        }
        std::fwrite(row.data(), 1, padded, file);
    }
    std::fclose(file);
}

}  // namespace luma
