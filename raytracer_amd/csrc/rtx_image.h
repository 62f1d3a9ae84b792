// Image file I/O of the host side (see rtx_image.cpp).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace rtxh {

struct Image {
	int w = 0, h = 0;
	std::vector<uint8_t> rgb;  // w*h*3
};

// Loads an image as 8-bit RGB, vertically flipped like the reference's Texture constructor.
bool load_texture_file(const std::string& path, Image& img);

// JPEG (rtx_jpeg.cpp): baseline or progressive, to top-down RGB.
bool load_jpeg(const std::vector<uint8_t>& file, Image& img);

// The other containers of the reference's loader (rtx_image_formats.cpp), each to top-down RGB; is_* tests the
// signature alone.
bool is_bmp(const std::vector<uint8_t>& file);
bool is_gif(const std::vector<uint8_t>& file);
bool is_psd(const std::vector<uint8_t>& file);
bool is_pic(const std::vector<uint8_t>& file);
bool is_hdr(const std::vector<uint8_t>& file);
bool load_bmp(const std::vector<uint8_t>& file, Image& img);
bool load_gif(const std::vector<uint8_t>& file, Image& img);
bool load_psd(const std::vector<uint8_t>& file, Image& img);
bool load_pic(const std::vector<uint8_t>& file, Image& img);
bool load_hdr(const std::vector<uint8_t>& file, Image& img);

// Writes an 8-bit RGB PNG (top row first).
bool write_png(const std::string& path, int w, int h, const uint8_t* rgb);

}  // namespace rtxh
