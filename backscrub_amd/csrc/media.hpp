// media.hpp — decoded background media (media.cpp): packed 8-bit BGR frames, as cv::imread / cv::VideoCapture hand them to the reference
// (/root/reference/app/background.cc:126-176).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace bsx {

struct Media {
  int width = 0, height = 0;
  double fps = 0;                                   // 0 for a still image
  std::vector<std::vector<uint8_t>> frames;         // [n][height][width][3] BGR
};

// GIF87a/89a, PNG (non-interlaced), JPEG (Huffman, sequential / progressive, 8-bit), binary PPM.  false + reason otherwise (WebM: no VP8/VP9
// decoder in this build).
bool media_load(const std::string& path, Media* m, std::string* err);
// jpeg.cpp: false with *err empty when the bytes are not a JPEG stream at all
bool decode_jpeg(const std::vector<uint8_t>& file, Media* m, std::string* err);

}  // namespace bsx
