// FLAC decoder (host code, no kernels): the data loader's native audio reader.
//
// Reference: BreverDataset.load_file reads `audio/NNNNN_<source>.flac` members with the
// third-party `soundfile` / libsndfile (brever/data.py:259-268) and `torchaudio.info`
// (data.py:143) -- both absent from this image. FLAC is lossless and fully specified (RFC 9639):
// this is an own decoder of the subset a lossless PCM archive uses -- STREAMINFO, fixed and
// variable block sizes, 1-8 channels with left/side, side/right and mid/side decorrelation,
// 4..32 bits per sample, CONSTANT / VERBATIM / FIXED (order 0-4) / LPC (order 1-32) subframes,
// wasted bits, partitioned Rice residuals with 4- and 5-bit parameters and escape partitions,
// CRC-8 of frame headers and CRC-16 of frames verified. Samples come out as float32 scaled by
// 2^-(bps-1), which is what libsndfile returns for dtype float32. Pinned by the example streams
// of RFC 9639 appendix D and by round trips through the test suite's own encoder; bit-exactness
// against libsndfile itself is unpinned (no FLAC file and no libsndfile here).
//
// The decoder takes HOST pointers (it runs in DataLoader worker processes); everything else in
// this library takes device pointers.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/brever_hip.h"

namespace {

struct BitReader {
  const uint8_t* p; size_t n; size_t byte; int bit;       // bit: bits already consumed of p[byte]
  bool fail;
  BitReader(const uint8_t* data, size_t size) : p(data), n(size), byte(0), bit(0), fail(false) {}
  inline uint32_t read1() {
    if (byte >= n) { fail = true; return 0; }
    const uint32_t v = (p[byte] >> (7 - bit)) & 1u;
    if (++bit == 8) { bit = 0; ++byte; }
    return v;
  }
  uint64_t read(int bits) {                                // MSB first, bits <= 64
    uint64_t v = 0;
    while (bits > 0) {
      if (byte >= n) { fail = true; return 0; }
      const int avail = 8 - bit;
      const int take = bits < avail ? bits : avail;
      const uint32_t chunk = (p[byte] >> (avail - take)) & ((1u << take) - 1u);
      v = (v << take) | chunk;
      bit += take; bits -= take;
      if (bit == 8) { bit = 0; ++byte; }
    }
    return v;
  }
  int64_t read_signed(int bits) {
    if (bits == 0) return 0;
    const uint64_t v = read(bits);
    const uint64_t sign = 1ull << (bits - 1);
    return (int64_t)((v ^ sign) - sign);
  }
  uint32_t read_unary() {                                  // zeros before the next 1
    uint32_t q = 0;
    while (!fail && read1() == 0) ++q;
    return q;
  }
  void align() { if (bit) { bit = 0; ++byte; } }
};

uint8_t crc8(const uint8_t* d, size_t n) {                 // poly x^8 + x^2 + x + 1
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= d[i];
    for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));
  }
  return c;
}
uint16_t crc16(const uint8_t* d, size_t n) {               // poly x^16 + x^15 + x^2 + 1
  uint16_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= (uint16_t)(d[i] << 8);
    for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1));
  }
  return c;
}

struct StreamInfo { int rate, channels, bps; int64_t total; int max_block; };

// returns the offset of the first frame, or a negative error
int64_t parse_metadata(const uint8_t* d, size_t n, StreamInfo& si) {
  if (n < 4 + 4 + 34 || memcmp(d, "fLaC", 4) != 0) return -10;
  size_t off = 4;
  bool got = false;
  for (;;) {
    if (off + 4 > n) return -11;
    const bool last = d[off] & 0x80;
    const int type = d[off] & 0x7f;
    const size_t len = ((size_t)d[off + 1] << 16) | ((size_t)d[off + 2] << 8) | d[off + 3];
    off += 4;
    if (off + len > n) return -11;
    if (type == 0) {
      if (len < 34) return -12;
      const uint8_t* s = d + off;
      si.max_block = (s[2] << 8) | s[3];
      si.rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
      si.channels = ((s[12] >> 1) & 7) + 1;
      si.bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
      si.total = ((int64_t)(s[13] & 0xf) << 32) | ((int64_t)s[14] << 24) | (s[15] << 16) |
                 (s[16] << 8) | s[17];
      got = true;
    }
    off += len;
    if (last) break;
  }
  return got ? (int64_t)off : -12;
}

int decode_residual(BitReader& br, int32_t* res, int blocksize, int order) {
  const int method = (int)br.read(2);
  if (method > 1) return -30;
  const int pbits = method == 0 ? 4 : 5;
  const int esc = (1 << pbits) - 1;
  const int porder = (int)br.read(4);
  const int parts = 1 << porder;
  if (porder > 0 && ((blocksize >> porder) << porder) != blocksize) return -31;
  int pos = 0;
  for (int part = 0; part < parts; ++part) {
    int count = porder == 0 ? blocksize - order : (blocksize >> porder) - (part == 0 ? order : 0);
    if (count < 0) return -32;
    const int k = (int)br.read(pbits);
    if (k == esc) {
      const int nb = (int)br.read(5);
      for (int i = 0; i < count; ++i) res[pos++] = (int32_t)br.read_signed(nb);
    } else {
      for (int i = 0; i < count; ++i) {
        const uint32_t q = br.read_unary();
        const uint32_t u = (q << k) | (uint32_t)br.read(k);
        res[pos++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
      }
    }
    if (br.fail) return -33;
  }
  return 0;
}

int decode_subframe(BitReader& br, int64_t* out, int blocksize, int bps, std::vector<int32_t>& res) {
  if (br.read1()) return -20;                              // padding bit
  const int type = (int)br.read(6);
  int wasted = 0;
  if (br.read1()) wasted = (int)br.read_unary() + 1;
  bps -= wasted;
  if (bps < 1) return -21;
  if (type == 0) {                                         // CONSTANT
    const int64_t v = br.read_signed(bps);
    for (int i = 0; i < blocksize; ++i) out[i] = v;
  } else if (type == 1) {                                  // VERBATIM
    for (int i = 0; i < blocksize; ++i) out[i] = br.read_signed(bps);
  } else if (type >= 8 && type <= 12) {                    // FIXED, order type - 8
    const int order = type - 8;
    if (order > blocksize) return -22;
    for (int i = 0; i < order; ++i) out[i] = br.read_signed(bps);
    res.resize(blocksize);
    if (int r = decode_residual(br, res.data(), blocksize, order)) return r;
    static const int coef[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
    for (int i = order; i < blocksize; ++i) {
      int64_t pred = 0;
      for (int j = 0; j < order; ++j) pred += (int64_t)coef[order][j]*out[i - 1 - j];
      out[i] = pred + res[i - order];
    }
  } else if (type >= 32) {                                 // LPC, order type - 31
    const int order = type - 31;
    if (order > blocksize) return -22;
    for (int i = 0; i < order; ++i) out[i] = br.read_signed(bps);
    const int precision = (int)br.read(4) + 1;
    if (precision == 16) return -23;
    const int shift = (int)br.read_signed(5);
    if (shift < 0) return -24;
    int32_t coefs[32];
    for (int j = 0; j < order; ++j) coefs[j] = (int32_t)br.read_signed(precision);
    res.resize(blocksize);
    if (int r = decode_residual(br, res.data(), blocksize, order)) return r;
    for (int i = order; i < blocksize; ++i) {
      int64_t pred = 0;
      for (int j = 0; j < order; ++j) pred += (int64_t)coefs[j]*out[i - 1 - j];
      out[i] = (pred >> shift) + res[i - order];
    }
  } else {
    return -25;                                            // reserved subframe type
  }
  if (wasted)
    for (int i = 0; i < blocksize; ++i) out[i] = (int64_t)((uint64_t)out[i] << wasted);
  return br.fail ? -26 : 0;
}

}  // namespace

extern "C" {

int brv_flac_info(const uint8_t* data, int64_t size, int64_t* frames, int32_t* sample_rate,
                  int32_t* channels, int32_t* bits_per_sample) {
  if (!data || size < 0) return -1;
  StreamInfo si;
  const int64_t off = parse_metadata(data, (size_t)size, si);
  if (off < 0) return (int)off;
  if (frames) *frames = si.total;
  if (sample_rate) *sample_rate = si.rate;
  if (channels) *channels = si.channels;
  if (bits_per_sample) *bits_per_sample = si.bps;
  return 0;
}

int64_t brv_flac_decode(const uint8_t* data, int64_t size, float* out, int64_t capacity_frames) {
  if (!data || !out || size < 0) return -1;
  StreamInfo si;
  int64_t off = parse_metadata(data, (size_t)size, si);
  if (off < 0) return off;
  const int C = si.channels;
  static const int bps_table[8] = {0, 8, 12, -1, 16, 20, 24, 32};
  std::vector<int64_t> chan((size_t)C*65536);
  std::vector<int32_t> res;
  int64_t done = 0;
  const double scale = 1.0/(double)(1ull << (si.bps - 1));
  while ((size_t)off + 2 <= (size_t)size) {
    const uint8_t* f = data + off;
    if (f[0] != 0xff || (f[1] & 0xfe) != 0xf8) return -40;              // sync + reserved bit
    BitReader br(f, (size_t)size - (size_t)off);
    br.read(16);
    const int bs_code = (int)br.read(4), sr_code = (int)br.read(4);
    const int ch_code = (int)br.read(4), ss_code = (int)br.read(3);
    if (br.read1()) return -41;
    // UTF-8 style coded frame / sample number
    {
      const uint32_t first = (uint32_t)br.read(8);
      int extra = 0;
      if (first & 0x80) { uint32_t m = 0x40; while (first & m) { ++extra; m >>= 1; } if (extra < 1 || extra > 6) return -42; }
      for (int i = 0; i < extra; ++i) if (((uint32_t)br.read(8) & 0xc0) != 0x80) return -42;
    }
    int blocksize;
    if (bs_code == 0) return -43;
    else if (bs_code == 1) blocksize = 192;
    else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
    else if (bs_code == 6) blocksize = (int)br.read(8) + 1;
    else if (bs_code == 7) blocksize = (int)br.read(16) + 1;
    else blocksize = 256 << (bs_code - 8);
    if (sr_code == 12) br.read(8); else if (sr_code == 13 || sr_code == 14) br.read(16);
    else if (sr_code == 15) return -44;
    if (br.fail || br.bit != 0) return -45;
    const size_t hdr_len = br.byte;
    if (crc8(f, hdr_len) != (uint8_t)br.read(8)) return -46;
    int bps = ss_code == 0 ? si.bps : bps_table[ss_code];
    if (bps <= 0 || bps != si.bps) return -47;
    int nch, mode = 0;                                    // 1 left/side, 2 side/right, 3 mid/side
    if (ch_code < 8) nch = ch_code + 1;
    else if (ch_code <= 10) { nch = 2; mode = ch_code - 7; }
    else return -48;
    if (nch != C || blocksize > 65536) return -49;
    for (int c = 0; c < nch; ++c) {
      const bool side = (mode == 1 && c == 1) || (mode == 2 && c == 0) || (mode == 3 && c == 1);
      if (int r = decode_subframe(br, chan.data() + (size_t)c*65536, blocksize, bps + (side ? 1 : 0), res))
        return r;
    }
    br.align();
    const size_t body = br.byte;
    const uint16_t want = (uint16_t)br.read(16);
    if (br.fail) return -50;
    if (crc16(f, body) != want) return -51;
    int64_t* a = chan.data(); int64_t* b = chan.data() + 65536;
    if (mode == 1) for (int i = 0; i < blocksize; ++i) b[i] = a[i] - b[i];
    else if (mode == 2) for (int i = 0; i < blocksize; ++i) a[i] = a[i] + b[i];
    else if (mode == 3)
      for (int i = 0; i < blocksize; ++i) {
        const int64_t mid = a[i]*2 + (b[i] & 1), s = b[i];
        a[i] = (mid + s) >> 1; b[i] = (mid - s) >> 1;
      }
    for (int i = 0; i < blocksize; ++i) {
      if (done + i >= capacity_frames) break;
      for (int c = 0; c < C; ++c)
        out[(done + i)*C + c] = (float)((double)chan[(size_t)c*65536 + i]*scale);
    }
    done += blocksize;
    off += (int64_t)br.byte;
    if (si.total > 0 && done >= si.total) break;
  }
  if (si.total > 0 && done > si.total) done = si.total;
  return done;
}

}  // extern "C"
