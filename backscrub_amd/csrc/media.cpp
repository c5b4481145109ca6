// media.cpp — still / animated background decoding for the background source (bsx_background_*, live.cpp).
//
// The reference opens its background through OpenCV (cv::VideoCapture with the ffmpeg backend, then cv::imread:
// /root/reference/app/background.cc:126-176).  Neither library exists in this image, so the formats the reference's own
// `backgrounds/` directory uses and that can be decoded without a codec library are implemented here from their public
// specifications: GIF87a/89a (LZW, interlace, local palettes, transparency, disposal methods — `animated.gif`), PNG (8-bit
// grey / RGB / palette / grey+alpha / RGBA, non-interlaced, inflate through zlib — `*.png`), JPEG (jpeg.cpp: sequential and progressive
// Huffman, libjpeg's default reconstruction — `*.jpg`) and binary PPM.  WebM is reported as unsupported (VP8/VP9: the caller hands
// decoded frames to bsx_background_from_frames instead).
// Output convention = cv::imread(IMREAD_COLOR) / VideoCapture with CONVERT_RGB: packed 8-bit BGR, alpha dropped.
#include "media.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstring>

namespace bsx {
namespace {

bool read_file(const std::string& path, std::vector<uint8_t>* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n <= 0 || n > (1l << 30)) { fclose(f); return false; }
  out->resize((size_t)n);
  const bool ok = fread(out->data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok;
}

// ---- GIF ------------------------------------------------------------------------------------------------------------------
struct Reader {
  const uint8_t* p; size_t n, at = 0; bool ok = true;
  uint8_t u8() { if (at >= n) { ok = false; return 0; } return p[at++]; }
  unsigned u16() { const unsigned a = u8(); return a | (u8() << 8); }
  bool skip(size_t k) { if (at + k > n) { ok = false; at = n; return false; } at += k; return true; }
};

// LZW image data: sub-blocks → codes (LSB first, variable width) → palette indices
bool gif_lzw(Reader& r, int min_code, std::vector<uint8_t>* idx, size_t want) {
  if (min_code < 2 || min_code > 8) return false;
  std::vector<uint8_t> data;
  for (;;) { const unsigned len = r.u8(); if (!r.ok) return false; if (!len) break; if (r.at + len > r.n) return false; data.insert(data.end(), r.p + r.at, r.p + r.at + len); r.at += len; }
  const int clear = 1 << min_code, eoi = clear + 1;
  int width = min_code + 1, next = eoi + 1, prev = -1;
  std::vector<int> prefix(4096, -1);
  std::vector<uint8_t> suffix(4096, 0), stack;
  for (int i = 0; i < clear; i++) suffix[i] = (uint8_t)i;
  uint32_t acc = 0; int bits = 0; size_t pos = 0;
  idx->clear(); idx->reserve(want);
  while (idx->size() < want) {
    while (bits < width && pos < data.size()) { acc |= (uint32_t)data[pos++] << bits; bits += 8; }
    if (bits < width) break;
    int code = (int)(acc & ((1u << width) - 1)); acc >>= width; bits -= width;
    if (code == clear) { width = min_code + 1; next = eoi + 1; prev = -1; continue; }
    if (code == eoi) break;
    int cur = code;
    stack.clear();
    if (prev < 0) { if (code >= clear) return false; idx->push_back((uint8_t)code); prev = code; continue; }
    if (code >= next) {                       // the KwKwK case: the code being defined right now
      if (code != next) return false;
      int t = prev; while (t >= clear) t = prefix[t];
      stack.push_back(suffix[t]); cur = prev;
    }
    while (cur >= clear) { stack.push_back(suffix[cur]); cur = prefix[cur]; if (stack.size() > 4096) return false; }
    stack.push_back(suffix[cur]);
    const uint8_t first = stack.back();
    for (size_t k = stack.size(); k-- > 0 && idx->size() < want;) idx->push_back(stack[k]);
    if (next < 4096) { prefix[next] = prev; suffix[next] = first; next++; if (next == (1 << width) && width < 12) width++; }
    prev = code;
  }
  return true;
}

bool decode_gif(const std::vector<uint8_t>& file, Media* m, std::string* err) {
  Reader r{file.data(), file.size()};
  if (file.size() < 13 || memcmp(file.data(), "GIF8", 4)) return false;
  r.at = 6;
  const int W = (int)r.u16(), H = (int)r.u16();
  const uint8_t flags = r.u8(); const uint8_t bg_index = r.u8(); r.u8();
  if (W <= 0 || H <= 0 || (long)W * H > (1l << 26)) { *err = "GIF: bad screen size"; return false; }
  std::vector<uint8_t> gpal(768, 0);
  int gpal_n = 0;
  if (flags & 0x80) { gpal_n = 2 << (flags & 7); if (!r.skip(0)) return false; if (r.at + 3 * (size_t)gpal_n > r.n) { *err = "GIF: truncated palette"; return false; } memcpy(gpal.data(), r.p + r.at, 3 * (size_t)gpal_n); r.at += 3 * (size_t)gpal_n; }
  // Canvas in RGB.  cv::VideoCapture reads GIFs through libavcodec's decoder, whose compositing rules are followed here: the logical
  // screen starts as the background colour when the first image carries no transparent index (and a global palette exists), else as
  // transparent; "restore to background" fills the frame's rectangle the same way.  libavcodec's transparent pixel is "transparent WHITE"
  // (0x00ffffff, chosen there so that formats without alpha show white, not black), so dropping alpha (BGRA→BGR) leaves 255,255,255.
  uint8_t bgc[3] = {0, 0, 0};
  if ((flags & 0x80) && bg_index < gpal_n) memcpy(bgc, &gpal[3 * bg_index], 3);
  std::vector<uint8_t> canvas((size_t)W * H * 3, 255), restore;
  auto fill = [&](int x0, int y0, int w, int h, const uint8_t* c) {
    for (int y = y0; y < y0 + h; y++) for (int x = x0; x < x0 + w; x++) memcpy(&canvas[((size_t)y * W + x) * 3], c, 3);
  };
  static const uint8_t clear[3] = {255, 255, 255};
  int transparent = -1, disposal = 0, delay_cs = 0;
  long total_delay = 0;
  m->width = W; m->height = H; m->frames.clear();
  while (r.ok && r.at < r.n) {
    const uint8_t b = r.u8();
    if (b == 0x3B) break;
    if (b == 0x21) {                                        // extension
      const uint8_t label = r.u8();
      if (label == 0xF9) {                                  // graphic control
        const unsigned len = r.u8();
        if (len >= 4) { const uint8_t pf = r.u8(); delay_cs = (int)r.u16(); const uint8_t ti = r.u8(); disposal = (pf >> 2) & 7; transparent = (pf & 1) ? ti : -1; r.skip(len - 4); }
        else r.skip(len);
      }
      for (;;) { const unsigned len = r.u8(); if (!r.ok || !len) break; r.skip(len); }   // remaining sub-blocks
      continue;
    }
    if (b != 0x2C) { *err = "GIF: unexpected block"; return false; }
    const int fx = (int)r.u16(), fy = (int)r.u16(), fw = (int)r.u16(), fh = (int)r.u16();
    const uint8_t lf = r.u8();
    const uint8_t* pal = gpal.data();
    std::vector<uint8_t> lpal;
    if (lf & 0x80) { const int n = 2 << (lf & 7); if (r.at + 3 * (size_t)n > r.n) { *err = "GIF: truncated local palette"; return false; } lpal.assign(r.p + r.at, r.p + r.at + 3 * (size_t)n); lpal.resize(768, 0); r.at += 3 * (size_t)n; pal = lpal.data(); }
    const int min_code = r.u8();
    if (fw <= 0 || fh <= 0 || fx + fw > W || fy + fh > H) { *err = "GIF: frame outside the logical screen"; return false; }
    std::vector<uint8_t> idx;
    if (!gif_lzw(r, min_code, &idx, (size_t)fw * fh)) { *err = "GIF: corrupt LZW stream"; return false; }
    idx.resize((size_t)fw * fh, 0);
    if (m->frames.empty() && transparent < 0 && (flags & 0x80)) fill(0, 0, W, H, bgc);
    if (disposal == 3) restore = canvas;
    // interlace: rows arrive in 4 passes
    std::vector<int> rowmap(fh);
    if (lf & 0x40) { int k = 0; for (int s : {0, 4, 2, 1}) for (int y = s, step = (s == 0 ? 8 : s == 4 ? 8 : s == 2 ? 4 : 2); y < fh; y += step) rowmap[k++] = y; }
    else for (int y = 0; y < fh; y++) rowmap[y] = y;
    for (int k = 0; k < fh; k++) {
      const int y = rowmap[k];
      for (int x = 0; x < fw; x++) {
        const int c = idx[(size_t)k * fw + x];
        if (c == transparent) continue;
        uint8_t* d = &canvas[((size_t)(fy + y) * W + fx + x) * 3];
        d[0] = pal[3 * c]; d[1] = pal[3 * c + 1]; d[2] = pal[3 * c + 2];
      }
    }
    std::vector<uint8_t> bgr((size_t)W * H * 3);
    for (size_t i = 0; i < (size_t)W * H; i++) { bgr[3 * i] = canvas[3 * i + 2]; bgr[3 * i + 1] = canvas[3 * i + 1]; bgr[3 * i + 2] = canvas[3 * i]; }
    m->frames.push_back(std::move(bgr));
    total_delay += delay_cs < 2 ? 10 : delay_cs;            // browsers / ffmpeg play delays below 2 cs as 10 cs
    if (disposal == 2) fill(fx, fy, fw, fh, transparent >= 0 ? clear : bgc);
    else if (disposal == 3) canvas = restore;
    transparent = -1; disposal = 0; delay_cs = 0;
    if (m->frames.size() > 4096 || m->frames.size() * (size_t)W * H * 3 > (size_t)1 << 30) { *err = "GIF: animation larger than 1 GiB decoded"; return false; }
  }
  if (m->frames.empty()) { *err = "GIF: no image"; return false; }
  m->fps = m->frames.size() > 1 ? 100.0 * (double)m->frames.size() / (double)total_delay : 0.0;
  return true;
}

// ---- PNG ------------------------------------------------------------------------------------------------------------------
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
bool decode_png(const std::vector<uint8_t>& file, Media* m, std::string* err) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 13, 10, 26, 10};
  if (file.size() < 33 || memcmp(file.data(), sig, 8)) return false;
  size_t at = 8;
  int W = 0, H = 0, depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> pal(768, 0), z;
  while (at + 12 <= file.size()) {
    const uint32_t len = be32(&file[at]);
    const uint8_t* type = &file[at + 4];
    if (at + 12 + (size_t)len > file.size()) { *err = "PNG: truncated chunk"; return false; }
    const uint8_t* d = &file[at + 8];
    if (!memcmp(type, "IHDR", 4) && len >= 13) { W = (int)be32(d); H = (int)be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12]; }
    else if (!memcmp(type, "PLTE", 4)) memcpy(pal.data(), d, len < 768 ? len : 768);
    else if (!memcmp(type, "IDAT", 4)) z.insert(z.end(), d, d + len);
    else if (!memcmp(type, "IEND", 4)) break;
    at += 12 + (size_t)len;
  }
  if (W <= 0 || H <= 0 || (long)W * H > (1l << 26)) { *err = "PNG: bad size"; return false; }
  if (interlace) { *err = "PNG: Adam7-interlaced images are not supported"; return false; }
  const int cn = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!cn) { *err = "PNG: unknown colour type"; return false; }
  const bool sub_byte = depth == 1 || depth == 2 || depth == 4;
  if (!(depth == 8 || (depth == 16 && ctype != 3) || (sub_byte && (ctype == 0 || ctype == 3)))) { *err = "PNG: bad bit depth"; return false; }
  const int fbpp = depth * cn >= 8 ? depth * cn / 8 : 1;     // the filters' "corresponding byte" distance (PNG spec §9.2)
  const size_t stride = ((size_t)W * depth * cn + 7) / 8;
  std::vector<uint8_t> raw((stride + 1) * (size_t)H);
  uLongf out_n = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_n, z.data(), (uLong)z.size()) != Z_OK || out_n != raw.size()) { *err = "PNG: inflate failed"; return false; }
  std::vector<uint8_t> img(stride * (size_t)H);
  for (int y = 0; y < H; y++) {                               // un-filter (PNG spec §9): None, Sub, Up, Average, Paeth
    const uint8_t f = raw[(stride + 1) * y];
    const uint8_t* s = &raw[(stride + 1) * y + 1];
    uint8_t* o = &img[stride * y];
    const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= (size_t)fbpp ? o[i - fbpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)fbpp) ? up[i - fbpp] : 0;
      int v = s[i];
      if (f == 1) v += a;
      else if (f == 2) v += b;
      else if (f == 3) v += (a + b) >> 1;
      else if (f == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
      else if (f != 0) { *err = "PNG: bad filter"; return false; }
      o[i] = (uint8_t)v;
    }
  }
  std::vector<uint8_t> bgr((size_t)W * H * 3);
  if (depth != 8) {                                            // → one byte per sample: 16-bit keeps the high byte (libpng strip_16, as
    std::vector<uint8_t> wide((size_t)W * H * cn);            // cv::imread(IMREAD_COLOR) asks for); 1/2/4-bit grey scales to 0..255, palette indices stay
    for (int y = 0; y < H; y++)
      for (size_t i = 0; i < (size_t)W * cn; i++) {
        const uint8_t* row = &img[stride * y];
        uint8_t v;
        if (depth == 16) v = row[2 * i];
        else {
          const int per = 8 / depth, sh = (per - 1 - (int)(i % per)) * depth;
          v = (uint8_t)((row[i / per] >> sh) & ((1 << depth) - 1));
          if (ctype == 0) v = (uint8_t)(v * 255 / ((1 << depth) - 1));
        }
        wide[(size_t)y * W * cn + i] = v;
      }
    img.swap(wide);
  }
  for (size_t i = 0; i < (size_t)W * H; i++) {
    uint8_t r, g, b;
    const uint8_t* p = &img[i * cn];
    if (ctype == 0 || ctype == 4) r = g = b = p[0];
    else if (ctype == 3) { r = pal[3 * p[0]]; g = pal[3 * p[0] + 1]; b = pal[3 * p[0] + 2]; }
    else { r = p[0]; g = p[1]; b = p[2]; }
    bgr[3 * i] = b; bgr[3 * i + 1] = g; bgr[3 * i + 2] = r;
  }
  m->width = W; m->height = H; m->fps = 0; m->frames.clear(); m->frames.push_back(std::move(bgr));
  return true;
}

bool decode_ppm(const std::vector<uint8_t>& file, Media* m, std::string* err) {
  if (file.size() < 11 || file[0] != 'P' || file[1] != '6') return false;
  size_t at = 2; int vals[3], k = 0;
  while (k < 3 && at < file.size()) {
    while (at < file.size() && (file[at] == ' ' || file[at] == '\n' || file[at] == '\r' || file[at] == '\t')) at++;
    if (at < file.size() && file[at] == '#') { while (at < file.size() && file[at] != '\n') at++; continue; }
    int v = 0, dg = 0; while (at < file.size() && file[at] >= '0' && file[at] <= '9') { v = v * 10 + (file[at++] - '0'); dg++; }
    if (!dg) break;
    vals[k++] = v;
  }
  if (k != 3 || vals[2] != 255 || vals[0] <= 0 || vals[1] <= 0) { *err = "PPM: unsupported header"; return false; }
  at++;
  const size_t n = (size_t)vals[0] * vals[1];
  if (at + 3 * n > file.size()) { *err = "PPM: truncated"; return false; }
  std::vector<uint8_t> bgr(3 * n);
  for (size_t i = 0; i < n; i++) { bgr[3 * i] = file[at + 3 * i + 2]; bgr[3 * i + 1] = file[at + 3 * i + 1]; bgr[3 * i + 2] = file[at + 3 * i]; }
  m->width = vals[0]; m->height = vals[1]; m->fps = 0; m->frames.clear(); m->frames.push_back(std::move(bgr));
  return true;
}

}  // namespace

bool media_load(const std::string& path, Media* m, std::string* err) {
  std::vector<uint8_t> file;
  std::string e;
  if (!read_file(path, &file)) { if (err) *err = "cannot open: " + path; return false; }
  bool ok = false;
  try {
    ok = decode_gif(file, m, &e) || (e.empty() && decode_png(file, m, &e)) || (e.empty() && decode_jpeg(file, m, &e)) || (e.empty() && decode_ppm(file, m, &e));
  } catch (const std::exception& ex) { e = ex.what(); ok = false; }
  if (!ok) {
    if (e.empty()) e = (file.size() > 4 && file[0] == 0x1A && file[1] == 0x45) ? "WebM/Matroska is not decodable here (no codec library): pass decoded frames instead"
                     : "unrecognised media format";
    if (err) *err = e + " (" + path + ")";
  }
  return ok;
}

}  // namespace bsx
