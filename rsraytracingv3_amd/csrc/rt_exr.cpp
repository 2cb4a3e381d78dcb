// rt_exr.cpp — OpenEXR reader for the skybox (host code, no GPU work).
//
// Stands in for `Connection::load_image` (src/gpu/connection.rs:154-162): the reference decodes
// ./assets/table_mountain_1_8k.exr with the `image` crate 0.25.1 (feature "exr", which wraps the
// `exr` crate 1.72.0 — both only named in Cargo.lock, neither vendored) and converts to RGBA32F
// (`into_rgba32f`, alpha 1 where the file has none).  The asset itself is not in the reference
// checkout (.MISSING_LARGE_BLOBS), so this reader follows the published OpenEXR file layout and
// is exercised against files written by an independent Python writer (tests/exr_writer.py):
// PARITY UNPINNED — no file produced by another EXR implementation was available to read.
//
// Supported: scan-line and tiled parts (of a tiled part the full-resolution level — ONE_LEVEL, or level (0, 0) of a
// MIPMAP / RIPMAP file: the reference asks for `largest_resolution_level`), single-part and multi-part files (the first
// part that is not deep and has R, G and B: the reference's `first_valid_layer`), compression NONE / RLE / ZIPS / ZIP /
// PIZ / PXR24, channel types HALF / FLOAT / UINT, no subsampling; channels R, G, B (required) and A (optional).
// Rejected with a message: deep-only files, B44 / B44A (lossy table-driven codecs the `exr` crate reads: not restated
// here without a file to check against) and DWAA / DWAB (the `exr` crate rejects those too).
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt355.h"

namespace {

thread_local std::string g_error;

int exr_fail(const std::string &msg) {
  g_error = msg;
  return RT_ERR_INVALID_ARGUMENT;
}

struct Reader {
  const uint8_t *p, *end;
  bool ok = true;
  template <typename T>
  T get() {
    T v{};
    if (static_cast<size_t>(end - p) < sizeof(T)) {
      ok = false;
      p = end;
      return v;
    }
    std::memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    return v;
  }
  std::string cstr(size_t limit = 255) {
    std::string s;
    while (p < end && *p != 0 && s.size() <= limit) s.push_back(static_cast<char>(*p++));
    if (p >= end || *p != 0) {
      ok = false;
      return s;
    }
    ++p;
    return s;
  }
};

constexpr size_t kMaxChannels = 64;                 // more is a damaged or hostile header
constexpr size_t kMaxBlockBytes = size_t{1} << 30;  // decoded bytes of one scan-line block

struct Channel {
  std::string name;
  int type = 0;  // 0 UINT, 1 HALF, 2 FLOAT
  int bytes() const { return type == 1 ? 2 : 4; }
};

float half_to_float(uint16_t h) {
  const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 31u, man = h & 1023u, bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal half: normalise
      int e = -1;
      do {
        man <<= 1;
        ++e;
      } while (!(man & 1024u));
      bits = sign | static_cast<uint32_t>(127 - 15 - e) << 23 | (man & 1023u) << 13;
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | man << 13;
  } else {
    bits = sign | (exp + 112u) << 23 | man << 13;
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

// --- ZIP / RLE post-processing: delta predictor + even/odd byte de-interleave ---------------
void unpredict_and_interleave(std::vector<uint8_t> &t, uint8_t *out) {
  const size_t n = t.size();
  for (size_t i = 1; i < n; ++i) t[i] = static_cast<uint8_t>(t[i - 1] + t[i] - 128);
  const size_t half = (n + 1) / 2;
  for (size_t i = 0; i < half; ++i) {
    out[2 * i] = t[i];
    if (2 * i + 1 < n) out[2 * i + 1] = t[half + i];
  }
}

bool rle_decode(const uint8_t *in, size_t n_in, std::vector<uint8_t> &out) {
  size_t o = 0;
  const size_t n_out = out.size();
  size_t i = 0;
  while (i < n_in) {
    const int c = static_cast<int8_t>(in[i++]);
    if (c < 0) {
      const size_t count = static_cast<size_t>(-c);
      if (i + count > n_in || o + count > n_out) return false;
      std::memcpy(&out[o], in + i, count);
      i += count;
      o += count;
    } else {
      const size_t count = static_cast<size_t>(c) + 1;
      if (i >= n_in || o + count > n_out) return false;
      std::memset(&out[o], in[i++], count);
      o += count;
    }
  }
  return o == n_out;
}

// --- PIZ: Huffman ------------------------------------------------------------------------------
constexpr int kHufEncSize = (1 << 16) + 1;
constexpr int kHufDecBits = 14;
constexpr int kShortZeroRun = 59, kLongZeroRun = 63, kShortestLongRun = 2 + kLongZeroRun - kShortZeroRun;

struct BitReader {
  const uint8_t *p, *end;
  uint64_t c = 0;
  int lc = 0;
  bool ok = true;
  uint32_t bits(int n) {
    while (lc < n) {
      if (p >= end) {
        ok = false;
        c <<= 8;
      } else {
        c = (c << 8) | *p++;
      }
      lc += 8;
    }
    lc -= n;
    return static_cast<uint32_t>((c >> lc) & ((1ull << n) - 1ull));
  }
};

struct LongCode {
  uint64_t code;
  int len;
  int sym;
};

bool huf_uncompress(const uint8_t *in, size_t n_in, uint16_t *out, size_t n_out) {
  if (n_in == 0) return n_out == 0;
  if (n_in < 20) return false;
  uint32_t hdr[5];
  std::memcpy(hdr, in, 20);
  const uint32_t im = hdr[0], iM = hdr[1], n_bits = hdr[3];
  if (im >= static_cast<uint32_t>(kHufEncSize) || iM >= static_cast<uint32_t>(kHufEncSize) || im > iM) return false;
  const uint8_t *ptr = in + 20;
  if (static_cast<uint64_t>(n_in - 20) * 8u < n_bits) return false;
  // code lengths (6 bits each, zero runs packed)
  std::vector<uint64_t> hcode(kHufEncSize, 0);
  BitReader br{ptr, in + n_in};
  for (uint32_t s = im; s <= iM; ++s) {
    const uint32_t l = br.bits(6);
    hcode[s] = l;
    if (l == static_cast<uint32_t>(kLongZeroRun)) {
      uint32_t zerun = br.bits(8) + kShortestLongRun;
      if (s + zerun > iM + 1) return false;
      while (zerun--) hcode[s++] = 0;
      --s;
    } else if (l >= static_cast<uint32_t>(kShortZeroRun)) {
      uint32_t zerun = l - kShortZeroRun + 2;
      if (s + zerun > iM + 1) return false;
      while (zerun--) hcode[s++] = 0;
      --s;
    }
  }
  if (!br.ok) return false;
  // the table is followed by the code stream, which starts on the next byte
  const uint8_t *data = br.p;
  if (static_cast<size_t>(data - ptr) != hdr[2]) return false;  // the header's table length must agree
  // canonical codes: shortest codes have the numerically largest prefixes
  uint64_t n[59] = {0};
  for (int i = 0; i < kHufEncSize; ++i) {
    if (hcode[i] > 58) return false;
    n[hcode[i]] += 1;
  }
  uint64_t c = 0;
  for (int i = 58; i > 0; --i) {
    const uint64_t nc = (c + n[i]) >> 1;
    n[i] = c;
    c = nc;
  }
  // decoding tables: direct lookup for codes of up to 14 bits, a list for the longer ones
  std::vector<int32_t> fast(1u << kHufDecBits, -1);  // sym | len << 24, -1 = none
  std::vector<LongCode> longs;
  for (uint32_t s = im; s <= iM; ++s) {
    const int l = static_cast<int>(hcode[s]);
    if (l == 0) continue;
    const uint64_t code = n[l]++;
    if (code >> l) return false;
    if (l <= kHufDecBits) {
      const uint32_t lo = static_cast<uint32_t>(code << (kHufDecBits - l));
      const uint32_t cnt = 1u << (kHufDecBits - l);
      for (uint32_t k = 0; k < cnt; ++k) {
        if (fast[lo + k] != -1) return false;
        fast[lo + k] = static_cast<int32_t>(s) | (l << 24);
      }
    } else {
      longs.push_back(LongCode{code, l, static_cast<int>(s)});
    }
  }
  // decode n_bits bits; symbol iM is the run-length escape (8-bit repeat count of the previous symbol)
  const uint32_t rlc = iM;
  struct {
    const uint8_t *p, *end;
    unsigned __int128 c;  // codes are up to 58 bits long: 58 + 7 pending bits do not fit 64
    int lc;
  } d{data, in + n_in, 0, 0};
  uint64_t used = 0;
  size_t o = 0;
  auto peek = [&](int nb) -> uint64_t {  // next nb (<= 58) bits, zero-padded past the end
    while (d.lc < nb) {
      d.c = (d.c << 8) | (d.p < d.end ? *d.p : 0u);
      if (d.p < d.end) ++d.p;
      d.lc += 8;
    }
    return static_cast<uint64_t>(d.c >> (d.lc - nb)) & ((1ull << nb) - 1ull);
  };
  auto skip = [&](int nb) {
    d.lc -= nb;
    used += static_cast<uint64_t>(nb);
  };
  while (used < n_bits) {
    int sym = -1, len = 0;
    const int32_t f = fast[static_cast<uint32_t>(peek(kHufDecBits))];
    if (f != -1) {
      sym = f & 0xffffff;
      len = f >> 24;
    } else {
      for (const LongCode &lc : longs) {
        if (peek(lc.len) == lc.code) {
          sym = lc.sym;
          len = lc.len;
          break;
        }
      }
      if (sym < 0) return false;
    }
    if (used + static_cast<uint64_t>(len) > n_bits) return false;
    skip(len);
    if (static_cast<uint32_t>(sym) == rlc) {
      if (used + 8 > n_bits || o == 0) return false;
      const uint32_t count = static_cast<uint32_t>(peek(8));
      skip(8);
      if (o + count > n_out) return false;
      const uint16_t prev = out[o - 1];
      for (uint32_t k = 0; k < count; ++k) out[o++] = prev;
    } else {
      if (o >= n_out) return false;
      out[o++] = static_cast<uint16_t>(sym);
    }
  }
  return o == n_out;
}

// --- PIZ: wavelet --------------------------------------------------------------------------------
inline void wdec14(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
  const int ls = static_cast<int16_t>(l), hs = static_cast<int16_t>(h);
  const int ai = ls + (hs & 1) + (hs >> 1);
  a = static_cast<uint16_t>(static_cast<int16_t>(ai));
  b = static_cast<uint16_t>(static_cast<int16_t>(ai - hs));
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
  const int m = l, d = h;
  const int bb = (m - (d >> 1)) & 0xffff;
  const int aa = (d + bb - (1 << 15)) & 0xffff;
  b = static_cast<uint16_t>(bb);
  a = static_cast<uint16_t>(aa);
}

void wav2_decode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t mx) {
  const bool w14 = mx < (1 << 14);
  const int n = nx > ny ? ny : nx;
  int p = 1, p2;
  while (p <= n) p <<= 1;
  p >>= 1;
  p2 = p;
  p >>= 1;
  auto dec = [&](uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    if (w14) wdec14(l, h, a, b); else wdec16(l, h, a, b);
  };
  while (p >= 1) {
    // positions are element indices into `in` (signed: the end markers may be negative)
    int64_t py = 0;
    const int64_t ey = static_cast<int64_t>(oy) * (ny - p2);
    const int64_t oy1 = static_cast<int64_t>(oy) * p, oy2 = static_cast<int64_t>(oy) * p2;
    const int64_t ox1 = static_cast<int64_t>(ox) * p, ox2 = static_cast<int64_t>(ox) * p2;
    uint16_t i00, i01, i10, i11;
    for (; py <= ey; py += oy2) {
      int64_t px = py;
      const int64_t ex = py + static_cast<int64_t>(ox) * (nx - p2);
      for (; px <= ex; px += ox2) {
        const int64_t p01 = px + ox1, p10 = px + oy1, p11 = p10 + ox1;
        dec(in[px], in[p10], i00, i10);
        dec(in[p01], in[p11], i01, i11);
        dec(i00, i01, in[px], in[p01]);
        dec(i10, i11, in[p10], in[p11]);
      }
      if (nx & p) {
        const int64_t p10 = px + oy1;
        dec(in[px], in[p10], i00, in[p10]);
        in[px] = i00;
      }
    }
    if (ny & p) {
      int64_t px = py;
      const int64_t ex = py + static_cast<int64_t>(ox) * (nx - p2);
      for (; px <= ex; px += ox2) {
        const int64_t p01 = px + ox1;
        dec(in[px], in[p01], i00, in[p01]);
        in[px] = i00;
      }
    }
    p2 = p;
    p >>= 1;
  }
}

bool piz_decode(const uint8_t *in, size_t n_in, const std::vector<Channel> &chans, int width, int lines, uint8_t *out) {
  Reader r{in, in + n_in};
  const uint16_t min_nz = r.get<uint16_t>(), max_nz = r.get<uint16_t>();
  std::vector<uint8_t> bitmap(8192, 0);
  if (max_nz >= 8192) return false;
  if (min_nz <= max_nz) {
    const size_t cnt = static_cast<size_t>(max_nz - min_nz) + 1;
    if (static_cast<size_t>(r.end - r.p) < cnt) return false;
    std::memcpy(&bitmap[min_nz], r.p, cnt);
    r.p += cnt;
  }
  std::vector<uint16_t> lut(65536, 0);
  uint32_t k = 0;
  for (uint32_t i = 0; i < 65536; ++i)
    if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = static_cast<uint16_t>(i);
  const uint16_t max_value = static_cast<uint16_t>(k - 1);
  const int32_t length = r.get<int32_t>();
  if (!r.ok || length < 0 || static_cast<size_t>(r.end - r.p) < static_cast<size_t>(length)) return false;
  size_t total = 0;
  for (const Channel &c : chans) total += static_cast<size_t>(width) * lines * (c.bytes() / 2);
  std::vector<uint16_t> tmp(total);
  if (!huf_uncompress(r.p, static_cast<size_t>(length), tmp.data(), total)) return false;
  // wavelet decode each 16-bit plane of each channel, then undo the value compaction
  size_t start = 0;
  std::vector<size_t> starts;
  for (const Channel &c : chans) {
    const int size = c.bytes() / 2;
    starts.push_back(start);
    for (int j = 0; j < size; ++j) wav2_decode(tmp.data() + start + j, width, size, lines, width * size, max_value);
    start += static_cast<size_t>(width) * lines * size;
  }
  for (uint16_t &v : tmp) v = lut[v];
  // back to scan-line order: per line, per channel, `width` samples
  uint8_t *o = out;
  std::vector<size_t> cursor = starts;
  for (int y = 0; y < lines; ++y) {
    for (size_t ci = 0; ci < chans.size(); ++ci) {
      const size_t nshort = static_cast<size_t>(width) * (chans[ci].bytes() / 2);
      std::memcpy(o, tmp.data() + cursor[ci], nshort * 2);
      cursor[ci] += nshort;
      o += nshort * 2;
    }
  }
  return true;
}

}  // namespace

namespace {

// One header of the file (a single-part file has one; a multi-part file a list ended by an empty header).
struct Part {
  std::vector<Channel> chans;
  int compression = -1, line_order = 0;
  int32_t data_win[4] = {0, 0, -1, -1}, disp_win[4] = {0, 0, -1, -1};
  bool have_data = false, have_disp = false;
  bool tiled = false, deep = false, have_tiles = false;
  uint32_t tile_x = 0, tile_y = 0;
  int level_mode = 0;  // 0 ONE_LEVEL, 1 MIPMAP, 2 RIPMAP: level (0, 0) is the full-resolution image in all three
  int64_t chunk_count = -1;
  std::string type;
  int idx[4] = {-1, -1, -1, -1};  // R G B A
  bool has_rgb() const { return idx[0] >= 0 && idx[1] >= 0 && idx[2] >= 0; }
};

// Reads one header's attributes up to its terminating null byte.  `empty` = the header had no attribute at all
// (the end of a multi-part file's header list).
int read_header(Reader &r, size_t name_limit, const std::string &bad, Part &part, bool &empty) {
  empty = true;
  for (;;) {
    if (r.p >= r.end) return exr_fail(bad + "truncated header");
    if (*r.p == 0) {
      ++r.p;
      break;
    }
    empty = false;
    const std::string name = r.cstr(name_limit), type = r.cstr(name_limit);
    const int32_t size = r.get<int32_t>();
    if (!r.ok || size < 0 || static_cast<size_t>(r.end - r.p) < static_cast<size_t>(size))
      return exr_fail(bad + "truncated attribute " + name);
    Reader a{r.p, r.p + size};
    r.p += size;
    if (name == "channels" && type == "chlist") {
      while (a.p < a.end && *a.p != 0) {
        Channel c;
        c.name = a.cstr(name_limit);
        c.type = a.get<int32_t>();
        (void)a.get<uint32_t>();  // pLinear + 3 reserved bytes
        const int32_t xs = a.get<int32_t>(), ys = a.get<int32_t>();
        if (!a.ok) return exr_fail(bad + "truncated channel list");
        if (c.type < 0 || c.type > 2) return exr_fail(bad + "unknown pixel type of channel " + c.name);
        if (xs != 1 || ys != 1) return exr_fail(bad + "subsampled channel " + c.name + " is not supported");
        if (part.chans.size() >= kMaxChannels) return exr_fail(bad + "more than 64 channels");
        part.chans.push_back(c);
      }
    } else if (name == "compression" && type == "compression") {
      part.compression = a.get<uint8_t>();
    } else if (name == "dataWindow" && type == "box2i") {
      for (int32_t &v : part.data_win) v = a.get<int32_t>();
      part.have_data = a.ok;
    } else if (name == "displayWindow" && type == "box2i") {
      for (int32_t &v : part.disp_win) v = a.get<int32_t>();
      part.have_disp = a.ok;
    } else if (name == "lineOrder" && type == "lineOrder") {
      part.line_order = a.get<uint8_t>();
    } else if (name == "tiles" && type == "tiledesc") {
      part.tile_x = a.get<uint32_t>();
      part.tile_y = a.get<uint32_t>();
      const uint8_t mode = a.get<uint8_t>();
      if (!a.ok) return exr_fail(bad + "truncated tile description");
      part.level_mode = mode & 15;
      part.have_tiles = true;
    } else if (name == "type" && type == "string") {
      part.type.assign(reinterpret_cast<const char *>(a.p), static_cast<size_t>(a.end - a.p));
    } else if (name == "chunkCount" && type == "int") {
      part.chunk_count = a.get<int32_t>();
      if (!a.ok || part.chunk_count < 0) return exr_fail(bad + "invalid chunkCount");
    }
  }
  for (size_t i = 0; i < part.chans.size(); ++i) {
    if (part.chans[i].name == "R") part.idx[0] = static_cast<int>(i);
    if (part.chans[i].name == "G") part.idx[1] = static_cast<int>(i);
    if (part.chans[i].name == "B") part.idx[2] = static_cast<int>(i);
    if (part.chans[i].name == "A") part.idx[3] = static_cast<int>(i);
  }
  return RT_OK;
}

// PXR24: the block's bytes are deflated after a per-line, per-channel split into byte planes (most significant
// first; a FLOAT keeps its upper 24 bits only) of running differences.  Output: scan-line order like every other
// decoder here (per line, per channel, `width` native samples).
bool pxr24_decode(const uint8_t *in, size_t n_in, const std::vector<Channel> &chans, int width, int lines, uint8_t *out,
                  std::vector<uint8_t> &tmp) {
  size_t packed_line = 0;
  for (const Channel &c : chans) packed_line += static_cast<size_t>(width) * (c.type == 1 ? 2 : (c.type == 2 ? 3 : 4));
  const size_t n_packed = packed_line * static_cast<size_t>(lines);
  tmp.resize(n_packed);
  uLongf got = static_cast<uLongf>(n_packed);
  if (uncompress(tmp.data(), &got, in, static_cast<uLong>(n_in)) != Z_OK || got != n_packed) return false;
  const uint8_t *p = tmp.data();
  const size_t n = static_cast<size_t>(width);
  for (int y = 0; y < lines; ++y) {
    for (const Channel &c : chans) {
      uint32_t pixel = 0;
      if (c.type == 1) {
        for (size_t x = 0; x < n; ++x) {
          pixel += static_cast<uint32_t>(p[x]) << 8 | p[n + x];
          const uint16_t h = static_cast<uint16_t>(pixel);
          std::memcpy(out + 2 * x, &h, 2);
        }
        p += 2 * n;
        out += 2 * n;
      } else if (c.type == 2) {
        for (size_t x = 0; x < n; ++x) {
          pixel += static_cast<uint32_t>(p[x]) << 24 | static_cast<uint32_t>(p[n + x]) << 16 | static_cast<uint32_t>(p[2 * n + x]) << 8;
          std::memcpy(out + 4 * x, &pixel, 4);
        }
        p += 3 * n;
        out += 4 * n;
      } else {
        for (size_t x = 0; x < n; ++x) {
          pixel += static_cast<uint32_t>(p[x]) << 24 | static_cast<uint32_t>(p[n + x]) << 16 | static_cast<uint32_t>(p[2 * n + x]) << 8 |
                   p[3 * n + x];
          std::memcpy(out + 4 * x, &pixel, 4);
        }
        p += 4 * n;
        out += 4 * n;
      }
    }
  }
  return true;
}

}  // namespace

extern "C" {

const char *rt_exr_last_error(void) { return g_error.c_str(); }

void rt_exr_free(float *rgba32f) { delete[] rgba32f; }

// No C++ exception crosses the C boundary (rt355.h): allocation failures inside the reader become
// RT_ERR_OUT_OF_MEMORY, anything else RT_ERR_INVALID_ARGUMENT.
static int exr_read_impl(const char *path, uint32_t *out_width, uint32_t *out_height, float **out_rgba32f);

int rt_exr_read_rgba32f(const char *path, uint32_t *out_width, uint32_t *out_height, float **out_rgba32f) {
  if (!path || !out_width || !out_height || !out_rgba32f) return exr_fail("rt_exr_read_rgba32f: NULL argument");
  *out_rgba32f = nullptr;
  try {
    return exr_read_impl(path, out_width, out_height, out_rgba32f);
  } catch (const std::bad_alloc &) {
    try {
      g_error = "rt_exr_read_rgba32f: out of memory";
    } catch (...) {
    }
    return RT_ERR_OUT_OF_MEMORY;
  } catch (...) {
    try {
      g_error = "rt_exr_read_rgba32f: internal error while decoding";
    } catch (...) {
    }
    return RT_ERR_INVALID_ARGUMENT;
  }
}

static int exr_read_impl(const char *path, uint32_t *out_width, uint32_t *out_height, float **out_rgba32f) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return exr_fail(std::string("Can't find image: ") + path);  // connection.rs:156
  std::vector<uint8_t> file;
  {
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n > 0) {
      file.resize(static_cast<size_t>(n));
      if (std::fread(file.data(), 1, file.size(), f) != file.size()) file.clear();
    }
    std::fclose(f);
  }
  const std::string bad = std::string("Image invalid format: ") + path + ": ";  // connection.rs:159
  Reader r{file.data(), file.data() + file.size()};
  if (r.get<uint32_t>() != 20000630u) return exr_fail(bad + "not an OpenEXR file (magic)");
  const uint32_t version = r.get<uint32_t>();
  if ((version & 0xffu) != 2u) return exr_fail(bad + "unsupported OpenEXR version");
  const bool multipart = (version & 0x1000u) != 0;
  if ((version & 0x800u) && !multipart) return exr_fail(bad + "deep data is not supported");
  const size_t name_limit = (version & 0x400u) ? 255 : 31;
  // Headers.  The reference's reader takes the first layer that has R, G and B and is not deep
  // (`image` 0.25.1 codecs/openexr.rs: first_valid_layer); so does this one.
  std::vector<Part> parts;
  for (;;) {
    Part part;
    bool empty = false;
    const int hrc = read_header(r, name_limit, bad, part, empty);
    if (hrc != RT_OK) return hrc;
    if (multipart && empty) break;
    if (!multipart) {
      part.tiled = (version & 0x200u) != 0;
    } else {
      part.tiled = part.type == "tiledimage";
      part.deep = part.type == "deepscanline" || part.type == "deeptile";
      if (!part.tiled && !part.deep && part.type != "scanlineimage") return exr_fail(bad + "part of unknown type '" + part.type + "'");
      if (part.chunk_count < 0) return exr_fail(bad + "multi-part header without chunkCount");
    }
    parts.push_back(std::move(part));
    if (!multipart) break;
    if (parts.size() > 1024) return exr_fail(bad + "more than 1024 parts");
  }
  size_t chosen = parts.size();
  for (size_t i = 0; i < parts.size() && chosen == parts.size(); ++i)
    if (!parts[i].deep && parts[i].has_rgb()) chosen = i;
  if (chosen == parts.size()) {
    if (!multipart && parts[0].chans.empty()) return exr_fail(bad + "required header attribute missing");
    return exr_fail(bad + (multipart ? "no part with R, G, B channels" : "no R, G, B channels"));
  }
  const Part &part = parts[chosen];
  const std::vector<Channel> &chans = part.chans;
  const int compression = part.compression;
  const int32_t *data_win = part.data_win, *disp_win = part.disp_win;
  const int *idx = part.idx;
  if (chans.empty() || compression < 0 || !part.have_data || !part.have_disp || (part.tiled && !part.have_tiles))
    return exr_fail(bad + "required header attribute missing");
  static const char *kNames[] = {"NONE", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB"};
  int lines_per_block;
  switch (compression) {
    case 0: case 1: case 2: lines_per_block = 1; break;
    case 3: case 5: lines_per_block = 16; break;
    case 4: lines_per_block = 32; break;
    default:
      return exr_fail(bad + "compression " + (compression < 10 ? kNames[compression] : "?") + " is not supported "
                            "(NONE, RLE, ZIPS, ZIP, PIZ, PXR24 are)");
  }
  // (lineOrder is not needed: every chunk carries its own coordinates; the offset table makes the order irrelevant)
  const int64_t dw = static_cast<int64_t>(data_win[2]) - data_win[0] + 1, dh = static_cast<int64_t>(data_win[3]) - data_win[1] + 1;
  const int64_t W = static_cast<int64_t>(disp_win[2]) - disp_win[0] + 1, H = static_cast<int64_t>(disp_win[3]) - disp_win[1] + 1;
  if (dw <= 0 || dh <= 0 || W <= 0 || H <= 0 || dw > 65536 || dh > 65536 || W > 65536 || H > 65536)
    return exr_fail(bad + "unreasonable data / display window");
  if (W * H > (1ll << 28) || dw * dh > (1ll << 28))  // 16k x 16k: 4 GiB of RGBA32F — far beyond any skybox
    return exr_fail(bad + "image larger than 2^28 pixels");
  size_t pixel_bytes = 0;
  for (const Channel &c : chans) pixel_bytes += static_cast<size_t>(c.bytes());
  // A block's decoded bytes are held twice (raw + tmp): bound them before anything is allocated.  The
  // largest sane block — 64 FLOAT channels would be absurd; 4 FLOAT channels x 65536 px x 32 lines is
  // 32 MiB — is far below this cap.
  int64_t block_w = dw, block_h = lines_per_block, tiles_x = 1, tiles_y = 1;
  if (part.tiled) {
    if (part.tile_x == 0 || part.tile_y == 0 || part.tile_x > 65536 || part.tile_y > 65536) return exr_fail(bad + "unreasonable tile size");
    if (part.level_mode > 2) return exr_fail(bad + "unknown tile level mode");
    block_w = std::min<int64_t>(part.tile_x, dw);
    block_h = std::min<int64_t>(part.tile_y, dh);
    tiles_x = (dw + part.tile_x - 1) / part.tile_x;
    tiles_y = (dh + part.tile_y - 1) / part.tile_y;
  }
  if (pixel_bytes * static_cast<size_t>(block_w) * static_cast<size_t>(block_h) > kMaxBlockBytes)
    return exr_fail(bad + (part.tiled ? "tile larger than 1 GiB" : "scan-line block larger than 1 GiB"));
  // Offset tables: one per part, in the parts' order.  Of a tiled part only the full-resolution level is read, whose
  // tiles come first in its table whatever the level mode.
  const size_t n_blocks = part.tiled ? static_cast<size_t>(tiles_x * tiles_y) : static_cast<size_t>((dh + lines_per_block - 1) / lines_per_block);
  for (size_t i = 0; i < chosen; ++i) {
    const uint64_t skip = static_cast<uint64_t>(parts[i].chunk_count) * 8u;  // multi-part: chunkCount is there (checked above)
    if (static_cast<uint64_t>(r.end - r.p) < skip) return exr_fail(bad + "truncated offset table");
    r.p += skip;
  }
  if (multipart && static_cast<uint64_t>(part.chunk_count) < n_blocks) return exr_fail(bad + "chunkCount smaller than the image needs");
  if (static_cast<size_t>(r.end - r.p) / 8 < n_blocks) return exr_fail(bad + "truncated offset table");
  std::vector<uint64_t> offsets(n_blocks);
  std::memcpy(offsets.data(), r.p, n_blocks * 8);

  std::unique_ptr<float[]> owner(new (std::nothrow) float[static_cast<size_t>(W) * H * 4]);
  float *rgba = owner.get();
  if (!rgba) {
    g_error = "rt_exr_read_rgba32f: out of memory";
    return RT_ERR_OUT_OF_MEMORY;
  }
  // outside the data window: transparent black
  std::memset(rgba, 0, static_cast<size_t>(W) * H * 16);
  std::vector<uint8_t> raw, tmp;
  for (size_t b = 0; b < n_blocks; ++b) {
    auto fail_block = [&](const char *why) { return exr_fail(bad + why); };
    if (offsets[b] > file.size() || file.size() - offsets[b] < 8) return fail_block("chunk offset outside the file");
    Reader c{file.data() + offsets[b], file.data() + file.size()};
    if (multipart && c.get<int32_t>() != static_cast<int32_t>(chosen)) return fail_block("chunk of another part in this part's offset table");
    int64_t bx0 = 0;  // first column of the block, relative to the data window
    int32_t y0;       // first line of the block, absolute
    int64_t width = dw;
    int lines;
    if (part.tiled) {
      const int32_t tx = c.get<int32_t>(), ty = c.get<int32_t>(), lx = c.get<int32_t>(), ly = c.get<int32_t>();
      if (!c.ok) return fail_block("truncated chunk");
      if (lx != 0 || ly != 0) return fail_block("tile of a reduced level where a full-resolution tile belongs");
      if (tx < 0 || ty < 0 || tx >= tiles_x || ty >= tiles_y) return fail_block("chunk with an invalid tile coordinate");
      bx0 = static_cast<int64_t>(tx) * part.tile_x;
      width = std::min<int64_t>(part.tile_x, dw - bx0);
      const int64_t by0 = static_cast<int64_t>(ty) * part.tile_y;
      y0 = static_cast<int32_t>(data_win[1] + by0);
      lines = static_cast<int>(std::min<int64_t>(part.tile_y, dh - by0));
    } else {
      y0 = c.get<int32_t>();
      if (y0 < data_win[1] || y0 > data_win[3] || (static_cast<int64_t>(y0) - data_win[1]) % lines_per_block != 0)
        return fail_block("chunk with an invalid scan-line coordinate");
      lines = static_cast<int>(std::min<int64_t>(lines_per_block, static_cast<int64_t>(data_win[3]) - y0 + 1));
    }
    const int32_t n_data = c.get<int32_t>();
    if (!c.ok || n_data < 0 || static_cast<size_t>(c.end - c.p) < static_cast<size_t>(n_data)) return fail_block("truncated chunk");
    const size_t line_bytes = pixel_bytes * static_cast<size_t>(width);
    const size_t n_raw = line_bytes * static_cast<size_t>(lines);
    raw.resize(n_raw);
    if (static_cast<size_t>(n_data) == n_raw) {
      std::memcpy(raw.data(), c.p, n_raw);  // stored uncompressed (NONE, or compression did not help)
    } else if (compression == 0) {
      return fail_block("uncompressed chunk of the wrong size");
    } else if (compression == 1) {
      tmp.resize(n_raw);
      if (!rle_decode(c.p, static_cast<size_t>(n_data), tmp)) return fail_block("corrupt RLE chunk");
      unpredict_and_interleave(tmp, raw.data());
    } else if (compression == 2 || compression == 3) {
      tmp.resize(n_raw);
      uLongf got = static_cast<uLongf>(n_raw);
      if (uncompress(tmp.data(), &got, c.p, static_cast<uLong>(n_data)) != Z_OK || got != n_raw)
        return fail_block("corrupt ZIP chunk");
      unpredict_and_interleave(tmp, raw.data());
    } else if (compression == 5) {
      if (!pxr24_decode(c.p, static_cast<size_t>(n_data), chans, static_cast<int>(width), lines, raw.data(), tmp))
        return fail_block("corrupt PXR24 chunk");
    } else if (!piz_decode(c.p, static_cast<size_t>(n_data), chans, static_cast<int>(width), lines, raw.data())) {
      return fail_block("corrupt PIZ chunk");
    }
    std::vector<size_t> chan_off;
    {
      size_t off = 0;
      for (const Channel &ch : chans) {
        chan_off.push_back(off);
        off += static_cast<size_t>(width) * ch.bytes();
      }
    }
    for (int l = 0; l < lines; ++l) {
      const int64_t y = static_cast<int64_t>(y0) + l - disp_win[1];  // row in the display window
      if (y < 0 || y >= H) continue;
      const uint8_t *line = raw.data() + line_bytes * static_cast<size_t>(l);
      for (int k = 0; k < 4; ++k) {
        float *dst = rgba + static_cast<size_t>(y) * W * 4 + k;
        if (idx[k] < 0) {  // no alpha channel: opaque (into_rgba32f)
          for (int64_t x = 0; x < width; ++x) {
            const int64_t X = bx0 + x + data_win[0] - disp_win[0];
            if (X >= 0 && X < W) dst[X * 4] = 1.0f;
          }
          continue;
        }
        const Channel &ch = chans[idx[k]];
        const uint8_t *src = line + chan_off[idx[k]];
        for (int64_t x = 0; x < width; ++x) {
          const int64_t X = bx0 + x + data_win[0] - disp_win[0];
          if (X < 0 || X >= W) continue;
          float v;
          if (ch.type == 1) {
            uint16_t h;
            std::memcpy(&h, src + 2 * x, 2);
            v = half_to_float(h);
          } else if (ch.type == 2) {
            std::memcpy(&v, src + 4 * x, 4);
          } else {
            uint32_t u;
            std::memcpy(&u, src + 4 * x, 4);
            v = static_cast<float>(u);
          }
          dst[X * 4] = v;
        }
      }
    }
  }
  *out_width = static_cast<uint32_t>(W);
  *out_height = static_cast<uint32_t>(H);
  *out_rgba32f = owner.release();
  return RT_OK;
}

int rt_set_skybox_exr(rt_context *ctx, const char *path) {
  if (!ctx) return RT_ERR_INVALID_ARGUMENT;
  uint32_t w = 0, h = 0;
  float *rgba = nullptr;
  const int rc = rt_exr_read_rgba32f(path, &w, &h, &rgba);  // catches every exception itself
  if (rc != RT_OK) return rc;
  const int rc2 = rt_set_skybox(ctx, w, h, rgba);
  rt_exr_free(rgba);
  return rc2;
}

}  // extern "C"
