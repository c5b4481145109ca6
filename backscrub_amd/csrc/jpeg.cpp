// jpeg.cpp — JPEG (ITU-T T.81) still-image decoding for the background source: what cv::imread(path) hands to load_background
// (/root/reference/app/background.cc:158-165) for `backgrounds/total_landscaping.jpg` (progressive, 4:2:0) and `screenshot.jpg`.
//
// OpenCV decodes JPEG through libjpeg(-turbo) with that library's defaults, and the pixels a JPEG decodes to are defined by those defaults,
// not by T.81 alone.  They are restated here from the library's published algorithm descriptions so that the result is the same bytes:
//   * Huffman entropy decoding, sequential (SOF0/SOF1) and progressive (SOF2: spectral selection + successive approximation), restart
//     intervals, interleaved and single-component scans;
//   * the "islow" inverse DCT — Loeffler-Ligtenberg-Moschytz, 13-bit constants, 2 extra bits between the passes — on de-quantised
//     coefficients, output through the wrap-around range table (index & 1023);
//   * "fancy" chroma up-sampling: the 3/4-1/4 triangle filter with the alternating 1/2 (h2v1) or 8/7 (h2v2) rounding biases, plain
//     replication where a component is at most two samples wide; edge rows/columns replicate;
//   * YCbCr → RGB with 16-bit fixed-point tables (1.402, 1.772, 0.71414, 0.34414; +0.5 folded into the blue/green tables);
//   * colour-space guess: JFIF → YCbCr; Adobe APP14 transform 0 → RGB, 1 → YCbCr; otherwise component ids 'R','G','B' → RGB;
//   * EXIF orientation (APP1, tag 0x0112) applied, as cv::imread does unless IMREAD_IGNORE_ORIENTATION.
// Not decoded (false + reason): arithmetic coding, lossless / hierarchical processes, 12-bit samples, CMYK/YCCK, sampling factors other than
// 1 or 2.  A progressive file whose scans never complete the first AC coefficients would additionally get libjpeg's inter-block smoothing;
// that case is decoded without it.
#include "media.hpp"

#include <cstring>

namespace bsx {
namespace {

struct Huff {
  bool present = false;
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  int maxcode[18], valptr[17], mincode[17];
  uint8_t look_n[256];
  uint8_t look_v[256];
  void build() {
    int code = 0, k = 0;
    memset(look_n, 0, sizeof(look_n));
    for (int l = 1; l <= 16; l++) {
      valptr[l] = k; mincode[l] = code;
      for (int i = 0; i < bits[l]; i++, k++, code++)
        if (l <= 8) { const int base = code << (8 - l); for (int j = 0; j < (1 << (8 - l)); j++) { look_n[base + j] = (uint8_t)l; look_v[base + j] = vals[k]; } }
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
  }
};

struct Comp {
  int id = 0, h = 1, v = 1, tq = 0;
  int wb = 0, hb = 0;                 // blocks per row / column of the coefficient array (padded to whole MCUs)
  int dw = 0, dh = 0;                 // down-sampled size in samples
  int td = 0, ta = 0;                 // tables of the current scan
  int pred = 0;
  std::vector<int16_t> coef;          // [hb][wb][64], natural order
  std::vector<uint8_t> plane;         // [hb*8][wb*8]
};

const uint8_t kZigzag[64 + 16] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};   // a corrupt run past 63 lands on the last coefficient

struct BitReader {
  const uint8_t* p; size_t n, at; uint32_t acc = 0; int cnt = 0; bool hit_marker = false;
  void reset() { acc = 0; cnt = 0; }
  void fill() {                                                // after a marker the stream reads as zero bits (what libjpeg does with a warning)
    while (cnt <= 24) {
      int b = 0;
      if (!hit_marker && at < n) {
        b = p[at];
        if (b == 0xFF) {
          size_t q = at + 1;
          while (q < n && p[q] == 0xFF) q++;                   // fill bytes
          if (q < n && p[q] == 0) at = q + 1;                  // stuffed zero: a data byte 0xFF
          else { hit_marker = true; b = 0; }
        } else at++;
      } else hit_marker = true;
      acc |= (uint32_t)b << (24 - cnt); cnt += 8;
    }
  }
  int peek8() { if (cnt < 16) fill(); return (int)(acc >> 24); }
  int get(int k) { if (!k) return 0; if (cnt < k) fill(); const int v = (int)(acc >> (32 - k)); acc <<= k; cnt -= k; return v; }
  int bit() { return get(1); }
  int decode(const Huff& h) {
    const int look = peek8();
    if (h.look_n[look]) { const int l = h.look_n[look]; acc <<= l; cnt -= l; return h.look_v[look]; }
    int code = get(8), l = 8;
    while (l < 16 && code > h.maxcode[l]) { code = (code << 1) | bit(); l++; }
    if (l == 16 && code > h.maxcode[16]) return 0;             // corrupt code
    if (h.maxcode[l] < 0 || code < h.mincode[l]) return 0;
    return h.vals[(h.valptr[l] + code - h.mincode[l]) & 255];
  }
  static int extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
  int receive_extend(int s) { return extend(get(s), s); }
};

// inverse DCT, "islow": 13-bit constants, 2 extra bits kept between the column and the row pass
inline int descale(long x, int n) { return (int)((x + (1l << (n - 1))) >> n); }
void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
  const long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819,
             F2562 = 20995, F3072 = 25172;
  int ws[64];
  for (int c = 0; c < 8; c++) {
    long d[8];
    for (int r = 0; r < 8; r++) d[r] = (long)in[8 * r + c] * q[8 * r + c];
    if (!(d[1] | d[2] | d[3] | d[4] | d[5] | d[6] | d[7])) { const int dc = (int)(d[0] * 4); for (int r = 0; r < 8; r++) ws[8 * r + c] = dc; continue; }
    long z2 = d[2], z3 = d[6];
    long z1 = (z2 + z3) * F0541;
    long tmp2 = z1 + z3 * -F1847, tmp3 = z1 + z2 * F0765;
    z2 = d[0]; z3 = d[4];
    long tmp0 = (z2 + z3) * 8192, tmp1 = (z2 - z3) * 8192;
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = d[7]; tmp1 = d[5]; tmp2 = d[3]; tmp3 = d[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    ws[0 + c] = descale(tmp10 + tmp3, 11); ws[56 + c] = descale(tmp10 - tmp3, 11);
    ws[8 + c] = descale(tmp11 + tmp2, 11); ws[48 + c] = descale(tmp11 - tmp2, 11);
    ws[16 + c] = descale(tmp12 + tmp1, 11); ws[40 + c] = descale(tmp12 - tmp1, 11);
    ws[24 + c] = descale(tmp13 + tmp0, 11); ws[32 + c] = descale(tmp13 - tmp0, 11);
  }
  auto limit = [](int x) -> uint8_t {                         // the post-IDCT range table indexed with (x & 1023), centre 128
    x &= 1023;
    return (uint8_t)(x < 128 ? x + 128 : x < 512 ? 255 : x < 896 ? 0 : x - 896);
  };
  for (int r = 0; r < 8; r++) {
    const int* w = &ws[8 * r];
    uint8_t* o = out + (size_t)r * stride;
    if (!(w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7])) { const uint8_t dc = limit(descale((long)w[0], 5)); for (int c = 0; c < 8; c++) o[c] = dc; continue; }
    long z2 = w[2], z3 = w[6];
    long z1 = (z2 + z3) * F0541;
    long tmp2 = z1 + z3 * -F1847, tmp3 = z1 + z2 * F0765;
    long tmp0 = ((long)w[0] + w[4]) * 8192, tmp1 = ((long)w[0] - w[4]) * 8192;
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    o[0] = limit(descale(tmp10 + tmp3, 18)); o[7] = limit(descale(tmp10 - tmp3, 18));
    o[1] = limit(descale(tmp11 + tmp2, 18)); o[6] = limit(descale(tmp11 - tmp2, 18));
    o[2] = limit(descale(tmp12 + tmp1, 18)); o[5] = limit(descale(tmp12 - tmp1, 18));
    o[3] = limit(descale(tmp13 + tmp0, 18)); o[4] = limit(descale(tmp13 - tmp0, 18));
  }
}

struct Decoder {
  const uint8_t* p; size_t n;
  std::string* err;
  int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1;
  bool progressive = false, have_sof = false, jfif = false, adobe = false;
  int adobe_transform = 0, orientation = 1, restart_interval = 0;
  uint16_t qt[4][64]; bool qt_ok[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  Comp comp[4];
  int mcux = 0, mcuy = 0;

  bool fail(const char* why) { *err = std::string("JPEG: ") + why; return false; }
  static unsigned be16(const uint8_t* q) { return ((unsigned)q[0] << 8) | q[1]; }

  bool parse_exif(const uint8_t* d, size_t len) {              // APP1 "Exif\0\0" + TIFF: IFD0 tag 0x0112
    if (len < 14 || memcmp(d, "Exif\0\0", 6)) return true;
    const uint8_t* t = d + 6; const size_t tn = len - 6;
    const bool le = t[0] == 'I' && t[1] == 'I';
    if (!le && !(t[0] == 'M' && t[1] == 'M')) return true;
    auto r16 = [&](size_t o) -> unsigned { return o + 2 <= tn ? (le ? t[o] | (t[o + 1] << 8) : (t[o] << 8) | t[o + 1]) : 0u; };
    auto r32 = [&](size_t o) -> size_t { return o + 4 <= tn ? (le ? (size_t)t[o] | ((size_t)t[o + 1] << 8) | ((size_t)t[o + 2] << 16) | ((size_t)t[o + 3] << 24)
                                                                  : ((size_t)t[o] << 24) | ((size_t)t[o + 1] << 16) | ((size_t)t[o + 2] << 8) | t[o + 3]) : 0u; };
    if (r16(2) != 42) return true;
    const size_t ifd = r32(4);
    const unsigned cnt = r16(ifd);
    for (unsigned i = 0; i < cnt && ifd + 2 + 12 * (size_t)(i + 1) <= tn; i++) {
      const size_t e = ifd + 2 + 12 * (size_t)i;
      if (r16(e) == 0x0112 && r16(e + 2) == 3) { const unsigned o = r16(e + 8); if (o >= 1 && o <= 8) orientation = (int)o; }
    }
    return true;
  }

  bool sof(const uint8_t* d, size_t len, int marker) {
    if (have_sof) return fail("more than one frame header");
    if (marker != 0xC0 && marker != 0xC1 && marker != 0xC2) return fail(marker == 0xC9 || marker == 0xCA ? "arithmetic coding is not supported" : "lossless / hierarchical processes are not supported");
    if (len < 6) return fail("truncated frame header");
    if (d[0] != 8) return fail("only 8-bit samples are supported");
    H = (int)be16(d + 1); W = (int)be16(d + 3); ncomp = d[5];
    if (W <= 0 || H <= 0 || (long)W * H > (1l << 26)) return fail("bad image size");
    if (ncomp == 4) return fail("CMYK / YCCK images are not supported");
    if ((ncomp != 1 && ncomp != 3) || len < 6 + 3 * (size_t)ncomp) return fail("unsupported component count");
    for (int i = 0; i < ncomp; i++) {
      Comp& c = comp[i];
      c.id = d[6 + 3 * i]; c.h = d[7 + 3 * i] >> 4; c.v = d[7 + 3 * i] & 15; c.tq = d[8 + 3 * i] & 3;
      if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2) return fail("sampling factors other than 1 or 2 are not supported");
      if (c.h > hmax) hmax = c.h;
      if (c.v > vmax) vmax = c.v;
    }
    if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }
    mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
    for (int i = 0; i < ncomp; i++) {
      Comp& c = comp[i];
      c.wb = mcux * c.h; c.hb = mcuy * c.v;
      c.dw = (W * c.h + hmax - 1) / hmax; c.dh = (H * c.v + vmax - 1) / vmax;
      c.coef.assign((size_t)c.wb * c.hb * 64, 0);
    }
    progressive = marker == 0xC2; have_sof = true;
    return true;
  }

  bool dht(const uint8_t* d, size_t len) {
    size_t at = 0;
    while (at + 17 <= len) {
      const int tc = d[at] >> 4, th = d[at] & 15;
      if (tc > 1 || th > 3) return fail("bad Huffman table id");
      Huff& h = tc ? ac[th] : dc[th];
      int total = 0;
      h.bits[0] = 0;
      for (int i = 1; i <= 16; i++) { h.bits[i] = d[at + i]; total += h.bits[i]; }
      if (total > 256 || at + 17 + (size_t)total > len) return fail("bad Huffman table");
      memset(h.vals, 0, sizeof(h.vals));
      memcpy(h.vals, d + at + 17, (size_t)total);
      h.present = true; h.build();
      at += 17 + (size_t)total;
    }
    return true;
  }

  bool dqt(const uint8_t* d, size_t len) {
    size_t at = 0;
    while (at < len) {
      const int pq = d[at] >> 4, tq = d[at] & 15;
      if (tq > 3 || pq > 1) return fail("bad quantisation table");
      const size_t need = pq ? 128 : 64;
      if (at + 1 + need > len) return fail("truncated quantisation table");
      for (int i = 0; i < 64; i++) qt[tq][kZigzag[i]] = pq ? (uint16_t)be16(d + at + 1 + 2 * i) : d[at + 1 + i];
      qt_ok[tq] = true;
      at += 1 + need;
    }
    return true;
  }

  // one scan: header at d, entropy-coded data from `pos`; returns the position of the marker that ends it
  bool scan(const uint8_t* d, size_t len, size_t* pos) {
    if (!have_sof) return fail("scan before the frame header");
    if (len < 1) return fail("truncated scan header");
    const int ns = d[0];
    if (ns < 1 || ns > ncomp || len < 4 + 2 * (size_t)ns) return fail("bad scan header");
    int ci[4];
    for (int i = 0; i < ns; i++) {
      int k = -1;
      for (int j = 0; j < ncomp; j++) if (comp[j].id == d[1 + 2 * i]) k = j;
      if (k < 0) return fail("scan names an unknown component");
      for (int j = 0; j < i; j++) if (ci[j] == k) return fail("scan names a component twice");
      ci[i] = k; comp[k].td = d[2 + 2 * i] >> 4; comp[k].ta = d[2 + 2 * i] & 15;
      if (comp[k].td > 3 || comp[k].ta > 3) return fail("bad table selector");
    }
    const int Ss = d[1 + 2 * ns], Se = d[2 + 2 * ns], Ah = d[3 + 2 * ns] >> 4, Al = d[3 + 2 * ns] & 15;
    if (progressive) {
      if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13 || (Ah && Ah != Al + 1)) return fail("bad progressive scan parameters");
    } else if (Ss != 0 || Se != 63 || Ah || Al) return fail("bad sequential scan parameters");
    for (int i = 0; i < ns; i++) {
      const Comp& c = comp[ci[i]];
      if ((!progressive || Ss == 0) && !(progressive && Ah) && !dc[c.td].present) return fail("scan uses an undefined DC table");
      if ((!progressive || Ss > 0) && !ac[c.ta].present) return fail("scan uses an undefined AC table");
    }
    BitReader br{p, n, *pos};
    int eobrun = 0, rst_left = restart_interval;
    for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
    // geometry: an interleaved scan walks MCUs; a single-component scan walks that component's blocks that lie inside the image
    const bool inter = ns > 1;
    const Comp& c0 = comp[ci[0]];
    const int ux = inter ? mcux : (c0.dw + 7) / 8, uy = inter ? mcuy : (c0.dh + 7) / 8;
    for (int my = 0; my < uy; my++)
      for (int mx = 0; mx < ux; mx++) {
        if (restart_interval && rst_left == 0) {
          // byte-align, expect RSTn
          br.reset(); br.hit_marker = false;
          size_t q = br.at;
          while (q + 1 < n && !(p[q] == 0xFF && p[q + 1] >= 0xD0 && p[q + 1] <= 0xD7)) { if (p[q] == 0xFF && p[q + 1] != 0 && p[q + 1] != 0xFF) break; q++; }
          if (q + 1 < n && p[q] == 0xFF && p[q + 1] >= 0xD0 && p[q + 1] <= 0xD7) br.at = q + 2; else br.hit_marker = true;
          for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
          eobrun = 0; rst_left = restart_interval;
        }
        for (int i = 0; i < ns; i++) {
          Comp& c = comp[ci[i]];
          const int bw = inter ? c.h : 1, bh = inter ? c.v : 1;
          for (int by = 0; by < bh; by++)
            for (int bx = 0; bx < bw; bx++) {
              const int X = (inter ? mx * c.h : mx) + bx, Y = (inter ? my * c.v : my) + by;
              int16_t* blk = &c.coef[((size_t)Y * c.wb + X) * 64];
              if (!progressive) {
                const int s = br.decode(dc[c.td]) & 15;
                c.pred += br.receive_extend(s);
                blk[0] = (int16_t)c.pred;
                for (int k = 1; k < 64;) {
                  const int rs = br.decode(ac[c.ta]), r = rs >> 4, sz = rs & 15;
                  if (!sz) { if (r != 15) break; k += 16; continue; }
                  k += r;
                  blk[kZigzag[k]] = (int16_t)br.receive_extend(sz);
                  k++;
                }
              } else if (Ss == 0) {
                if (!Ah) { const int s = br.decode(dc[c.td]) & 15; c.pred += br.receive_extend(s); blk[0] = (int16_t)(c.pred * (1 << Al)); }
                else if (br.bit()) blk[0] |= (int16_t)(1 << Al);
              } else if (!Ah) {                                  // AC first pass
                if (eobrun > 0) { eobrun--; continue; }
                for (int k = Ss; k <= Se;) {
                  const int rs = br.decode(ac[c.ta]), r = rs >> 4, sz = rs & 15;
                  if (!sz) {
                    if (r == 15) { k += 16; continue; }
                    eobrun = (1 << r) - 1; if (r) eobrun += br.get(r);
                    break;
                  }
                  k += r;
                  blk[kZigzag[k]] = (int16_t)(br.receive_extend(sz) * (1 << Al));
                  k++;
                }
              } else {                                           // AC refinement
                const int p1 = 1 << Al, m1 = -(1 << Al);
                int k = Ss;
                if (eobrun == 0) {
                  for (; k <= Se;) {
                    const int rs = br.decode(ac[c.ta]);
                    int r = rs >> 4; const int sz = rs & 15;
                    int val = 0;
                    if (sz) val = br.bit() ? p1 : m1;            // size must be 1
                    else if (r != 15) { eobrun = 1 << r; if (r) eobrun += br.get(r); break; }
                    for (; k <= Se; k++) {
                      int16_t* t = &blk[kZigzag[k]];
                      if (*t) { if (br.bit() && !(*t & p1)) *t = (int16_t)(*t >= 0 ? *t + p1 : *t + m1); }
                      else { if (--r < 0) break; }
                    }
                    if (val && k <= Se) blk[kZigzag[k]] = (int16_t)val;
                    k++;
                  }
                }
                if (eobrun > 0) {
                  for (; k <= Se; k++) { int16_t* t = &blk[kZigzag[k]]; if (*t && br.bit() && !(*t & p1)) *t = (int16_t)(*t >= 0 ? *t + p1 : *t + m1); }
                  eobrun--;
                }
              }
            }
        }
        if (restart_interval) rst_left--;
      }
    // advance to the next marker
    size_t q = br.at;
    if (q > 0 && br.hit_marker) { /* at sits on the 0xFF of the marker */ }
    while (q + 1 < n && !(p[q] == 0xFF && p[q + 1] != 0 && p[q + 1] != 0xFF && !(p[q + 1] >= 0xD0 && p[q + 1] <= 0xD7))) q++;
    *pos = q;
    return true;
  }

  bool parse() {
    if (n < 4 || p[0] != 0xFF || p[1] != 0xD8) return false;
    size_t at = 2;
    bool saw_scan = false;
    while (at + 4 <= n) {
      if (p[at] != 0xFF) { at++; continue; }
      const int m = p[at + 1];
      if (m == 0xFF) { at++; continue; }
      if (m == 0xD9) break;
      if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { at += 2; continue; }
      const size_t len = be16(p + at + 2);
      if (len < 2 || at + 2 + len > n) { if (saw_scan) break; return fail("truncated marker segment"); }
      const uint8_t* d = p + at + 4; const size_t dl = len - 2;
      at += 2 + len;
      if (m == 0xC4) { if (!dht(d, dl)) return false; }
      else if (m == 0xDB) { if (!dqt(d, dl)) return false; }
      else if (m == 0xDD) { if (dl < 2) return fail("truncated DRI"); restart_interval = (int)be16(d); }
      else if (m >= 0xC0 && m <= 0xCF && m != 0xC8 && m != 0xCC) { if (!sof(d, dl, m)) return false; }
      else if (m == 0xE0) { if (dl >= 5 && !memcmp(d, "JFIF", 5)) jfif = true; }
      else if (m == 0xE1) parse_exif(d, dl);
      else if (m == 0xEE) { if (dl >= 12 && !memcmp(d, "Adobe", 5)) { adobe = true; adobe_transform = d[11]; } }
      else if (m == 0xDA) { if (!scan(d, dl, &at)) return false; saw_scan = true; }
    }
    if (!have_sof || !saw_scan) return fail("no image data");
    return true;
  }

  void reconstruct() {
    for (int i = 0; i < ncomp; i++) {
      Comp& c = comp[i];
      static const uint16_t ones[64] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
      const uint16_t* q = qt_ok[c.tq] ? qt[c.tq] : ones;
      const int stride = c.wb * 8;
      c.plane.assign((size_t)stride * c.hb * 8, 0);
      for (int by = 0; by < c.hb; by++)
        for (int bx = 0; bx < c.wb; bx++) idct_islow(&c.coef[((size_t)by * c.wb + bx) * 64], q, &c.plane[(size_t)by * 8 * stride + (size_t)bx * 8], stride);
      c.coef.clear(); c.coef.shrink_to_fit();
    }
  }

  // component plane → full resolution [H][W] ("fancy" triangle filter where the library uses it)
  void upsample(const Comp& c, std::vector<uint8_t>* out) const {
    out->assign((size_t)W * H, 0);
    const int stride = c.wb * 8;
    const int fh = hmax / c.h, fv = vmax / c.v;
    auto row = [&](int y) -> const uint8_t* { return &c.plane[(size_t)(y < 0 ? 0 : y >= c.dh ? c.dh - 1 : y) * stride]; };
    if (fh == 1 && fv == 1) { for (int y = 0; y < H; y++) memcpy(&(*out)[(size_t)y * W], row(y), (size_t)W); return; }
    const bool fancy = c.dw > 2;
    std::vector<int> sum((size_t)c.dw);
    std::vector<uint8_t> wide((size_t)c.dw * 2 + 2);
    for (int y = 0; y < H; y++) {
      const int sy = y / fv;
      uint8_t* o = &(*out)[(size_t)y * W];
      if (fh == 2 && fv == 1) {
        const uint8_t* s = row(sy);
        if (!fancy) { for (int x = 0; x < W; x++) o[x] = s[x >> 1]; continue; }
        wide[0] = s[0]; wide[1] = (uint8_t)((s[0] * 3 + s[1] + 2) >> 2);
        for (int x = 1; x < c.dw - 1; x++) { wide[2 * x] = (uint8_t)((s[x] * 3 + s[x - 1] + 1) >> 2); wide[2 * x + 1] = (uint8_t)((s[x] * 3 + s[x + 1] + 2) >> 2); }
        wide[2 * (c.dw - 1)] = (uint8_t)((s[c.dw - 1] * 3 + s[c.dw - 2] + 1) >> 2); wide[2 * (c.dw - 1) + 1] = s[c.dw - 1];
        memcpy(o, wide.data(), (size_t)W);
      } else if (fh == 1 && fv == 2) {
        const uint8_t* s0 = row(sy); const uint8_t* s1 = row((y & 1) ? sy + 1 : sy - 1);
        const int bias = (y & 1) ? 2 : 1;
        for (int x = 0; x < W; x++) o[x] = (uint8_t)((s0[x] * 3 + s1[x] + bias) >> 2);
      } else {                                                  // h2v2
        const uint8_t* s0 = row(sy);
        if (!fancy) { for (int x = 0; x < W; x++) o[x] = s0[x >> 1]; continue; }
        const uint8_t* s1 = row((y & 1) ? sy + 1 : sy - 1);
        for (int x = 0; x < c.dw; x++) sum[x] = s0[x] * 3 + s1[x];
        wide[0] = (uint8_t)((sum[0] * 4 + 8) >> 4); wide[1] = (uint8_t)((sum[0] * 3 + sum[1] + 7) >> 4);
        for (int x = 1; x < c.dw - 1; x++) { wide[2 * x] = (uint8_t)((sum[x] * 3 + sum[x - 1] + 8) >> 4); wide[2 * x + 1] = (uint8_t)((sum[x] * 3 + sum[x + 1] + 7) >> 4); }
        wide[2 * (c.dw - 1)] = (uint8_t)((sum[c.dw - 1] * 3 + sum[c.dw - 2] + 8) >> 4); wide[2 * (c.dw - 1) + 1] = (uint8_t)((sum[c.dw - 1] * 4 + 7) >> 4);
        memcpy(o, wide.data(), (size_t)W);
      }
    }
  }

  bool to_bgr(std::vector<uint8_t>* bgr) {
    std::vector<uint8_t> pl[3];
    for (int i = 0; i < ncomp; i++) upsample(comp[i], &pl[i]);
    bgr->resize((size_t)W * H * 3);
    if (ncomp == 1) { for (size_t i = 0; i < (size_t)W * H; i++) { (*bgr)[3 * i] = (*bgr)[3 * i + 1] = (*bgr)[3 * i + 2] = pl[0][i]; } return true; }
    bool rgb = false;
    if (jfif) rgb = false;
    else if (adobe) rgb = adobe_transform == 0;
    else rgb = comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B';
    if (rgb) { for (size_t i = 0; i < (size_t)W * H; i++) { (*bgr)[3 * i] = pl[2][i]; (*bgr)[3 * i + 1] = pl[1][i]; (*bgr)[3 * i + 2] = pl[0][i]; } return true; }
    int crr[256], cbb[256]; long crg[256], cbg[256];
    for (int i = 0; i < 256; i++) {
      const long x = i - 128;
      crr[i] = (int)((91881 * x + 32768) >> 16); cbb[i] = (int)((116130 * x + 32768) >> 16);
      crg[i] = -46802 * x; cbg[i] = -22554 * x + 32768;
    }
    auto clamp = [](int v) -> uint8_t { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
    for (size_t i = 0; i < (size_t)W * H; i++) {
      const int y = pl[0][i], cb = pl[1][i], cr = pl[2][i];
      (*bgr)[3 * i] = clamp(y + cbb[cb]);
      (*bgr)[3 * i + 1] = clamp(y + (int)((cbg[cb] + crg[cr]) >> 16));
      (*bgr)[3 * i + 2] = clamp(y + crr[cr]);
    }
    return true;
  }
};

// EXIF orientations 2-8 (what cv::imread applies): mirror / rotate the decoded picture
void apply_orientation(int o, int* W, int* H, std::vector<uint8_t>* img) {
  if (o <= 1 || o > 8) return;
  const int w = *W, h = *H;
  const bool swap = o >= 5;
  const int ow = swap ? h : w, oh = swap ? w : h;
  std::vector<uint8_t> out((size_t)w * h * 3);
  for (int y = 0; y < oh; y++)
    for (int x = 0; x < ow; x++) {
      int sx, sy;
      switch (o) {
        case 2: sx = w - 1 - x; sy = y; break;                 // mirror horizontally
        case 3: sx = w - 1 - x; sy = h - 1 - y; break;         // rotate 180
        case 4: sx = x; sy = h - 1 - y; break;                 // mirror vertically
        case 5: sx = y; sy = x; break;                         // transpose
        case 6: sx = y; sy = h - 1 - x; break;                 // rotate 90 clockwise
        case 7: sx = w - 1 - y; sy = h - 1 - x; break;         // transverse
        default: sx = w - 1 - y; sy = x; break;                // 8: rotate 270 clockwise
      }
      memcpy(&out[((size_t)y * ow + x) * 3], &(*img)[((size_t)sy * w + sx) * 3], 3);
    }
  img->swap(out); *W = ow; *H = oh;
}

}  // namespace

bool decode_jpeg(const std::vector<uint8_t>& file, Media* m, std::string* err) {
  Decoder d;
  d.p = file.data(); d.n = file.size(); d.err = err;
  if (!d.parse()) return false;
  d.reconstruct();
  std::vector<uint8_t> bgr;
  if (!d.to_bgr(&bgr)) return false;
  int W = d.W, H = d.H;
  apply_orientation(d.orientation, &W, &H, &bgr);
  m->width = W; m->height = H; m->fps = 0; m->frames.clear(); m->frames.push_back(std::move(bgr));
  return true;
}

}  // namespace bsx
