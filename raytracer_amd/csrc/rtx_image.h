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

// Writes an 8-bit RGB PNG (top row first).
bool write_png(const std::string& path, int w, int h, const uint8_t* rgb);

}  // namespace rtxh
