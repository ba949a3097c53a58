// image.h -- luma::Image (image.h:9-22): PPM (P3) and 24-bit BMP writers over a 0x00RRGGBB frame.
#pragma once
#include <cstdint>
#include <string>

namespace luma {

class Image {
public:
    Image(std::int32_t width, std::int32_t height, bool allocate) noexcept;
    ~Image();
    Image(const Image&) = delete;
    Image& operator=(const Image&) = delete;

    void shadow_from(std::uint32_t* data) noexcept;          // alias the caller's frame (image.cpp:25-28)
    void export_to_ppm(const std::string& filepath) noexcept;
    void export_to_bmp(const std::string& filepath) noexcept;
    void export_to_pnm(const std::string& filepath) noexcept;    // binary P6, row-major (not in the reference)

private:
    const std::int32_t _width, _height;
    std::uint32_t* _data;
    bool _owned;
};

}  // namespace luma
