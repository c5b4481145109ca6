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

// GIF87a/89a, PNG (8-bit, non-interlaced), binary PPM.  false + reason otherwise (JPEG / WebM: no codec library in this build).
bool media_load(const std::string& path, Media* m, std::string* err);

}  // namespace bsx
