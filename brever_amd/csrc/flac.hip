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

// ---- encoder (host code): what scripts/test_model.py --output_dir writes ------------------------
// Reference: torchaudio.save(path, x, fs) with a .flac name (scripts/test_model.py:201-209). Mono,
// 16 bits per sample, blocks of 4096 samples; per block the FIXED predictor (order 0-4) with the
// smallest sum of |residual|, residuals Rice-coded in 2^p partitions (p <= 3 where the block divides)
// with the parameter that minimises each partition's length; CONSTANT subframes for constant blocks,
// VERBATIM when prediction does not pay. Lossless by construction; the decoder above reads it back
// bit for bit (tests/test_host.py). Which block size / predictor torchaudio's back end would have
// chosen does not change the decoded samples.
namespace {
struct BitWriter {
  std::vector<uint8_t>& out; uint64_t acc; int n;
  explicit BitWriter(std::vector<uint8_t>& o) : out(o), acc(0), n(0) {}
  void write(uint64_t v, int bits) {                       // MSB first, bits <= 32
    if (bits <= 0) return;
    acc = (acc << bits) | (v & ((bits >= 64) ? ~0ull : ((1ull << bits) - 1)));
    n += bits;
    while (n >= 8) { n -= 8; out.push_back((uint8_t)(acc >> n)); }
    acc &= (n ? ((1ull << n) - 1) : 0);
  }
  void unary(uint32_t q) { while (q >= 32) { write(0, 32); q -= 32; } write(1, (int)q + 1); }
  void align() { if (n) write(0, 8 - n); }
};
inline uint32_t zigzag(int64_t v) { return v >= 0 ? (uint32_t)(v << 1) : (uint32_t)(((-v) << 1) - 1); }
// bits of `count` residuals with Rice parameter k
inline uint64_t rice_bits(const int64_t* r, int count, int k) {
  uint64_t b = 0;
  for (int i = 0; i < count; ++i) b += (zigzag(r[i]) >> k) + 1 + k;
  return b;
}
}  // namespace

extern "C" int64_t brv_flac_encode16(const int16_t* pcm, int64_t frames, int32_t sample_rate, uint8_t* out,
                                     int64_t capacity) {
  if (!pcm || frames < 0 || sample_rate <= 0 || sample_rate >= (1 << 20)) return -1;
  const int BS = 4096;
  std::vector<uint8_t> buf;
  buf.reserve((size_t)frames*2 + 1024);
  const uint8_t magic[4] = {'f', 'L', 'a', 'C'};
  buf.insert(buf.end(), magic, magic + 4);
  {
    const uint8_t hdr[4] = {0x80, 0, 0, 34};               // last metadata block: STREAMINFO, 34 bytes
    buf.insert(buf.end(), hdr, hdr + 4);
    BitWriter bw(buf);
    bw.write(BS, 16); bw.write(BS, 16); bw.write(0, 24); bw.write(0, 24);
    bw.write((uint64_t)sample_rate, 20); bw.write(0, 3); bw.write(15, 5);
    bw.write((uint64_t)frames >> 32, 4); bw.write((uint64_t)frames & 0xffffffffull, 32);
    for (int i = 0; i < 4; ++i) bw.write(0, 32);           // MD5 unset
  }
  std::vector<int64_t> x(BS), res(BS);
  int64_t fno = 0;
  for (int64_t start = 0; start < frames; start += BS, ++fno) {
    const int n = (int)((frames - start) < BS ? (frames - start) : BS);
    for (int i = 0; i < n; ++i) x[i] = pcm[start + i];
    const size_t f0 = buf.size();
    BitWriter bw(buf);
    bw.write(0xfff8 >> 2, 14); bw.write(0, 1); bw.write(0, 1);     // sync, reserved, fixed block size
    bw.write(7, 4);                                        // 16-bit (block size - 1) at the end of the header
    bw.write(0, 4);                                        // sample rate: STREAMINFO
    bw.write(0, 4);                                        // mono
    bw.write(4, 3); bw.write(0, 1);                        // 16 bits per sample
    {                                                      // frame number, UTF-8 coded
      uint64_t v = (uint64_t)fno;
      if (v < 0x80) bw.write(v, 8);
      else {
        int extra = v < 0x800 ? 1 : v < 0x10000 ? 2 : v < 0x200000 ? 3 : v < 0x4000000 ? 4 : 5;
        bw.write(((0xff00u >> extra) & 0xff) | (v >> (6*extra)), 8);
        for (int e = extra - 1; e >= 0; --e) bw.write(0x80 | ((v >> (6*e)) & 0x3f), 8);
      }
    }
    bw.write((uint64_t)(n - 1), 16);
    buf.push_back(crc8(buf.data() + f0, buf.size() - f0));
    // ---- the subframe
    bool constant = true;
    for (int i = 1; i < n; ++i) if (x[i] != x[0]) { constant = false; break; }
    if (constant) {
      bw.write(0, 8); bw.write((uint64_t)x[0], 16);
    } else {
      // best fixed order by the sum of |residual| (orders above n - 1 are not available)
      int best_o = 0; uint64_t best_s = ~0ull;
      for (int o = 0; o <= 4 && o < n; ++o) {
        uint64_t sum = 0;
        for (int i = o; i < n; ++i) {
          int64_t r = x[i];
          if (o == 1) r = x[i] - x[i - 1];
          else if (o == 2) r = x[i] - 2*x[i - 1] + x[i - 2];
          else if (o == 3) r = x[i] - 3*x[i - 1] + 3*x[i - 2] - x[i - 3];
          else if (o == 4) r = x[i] - 4*x[i - 1] + 6*x[i - 2] - 4*x[i - 3] + x[i - 4];
          sum += (uint64_t)(r < 0 ? -r : r);
        }
        if (sum < best_s) { best_s = sum; best_o = o; }
      }
      const int o = best_o;
      for (int i = o; i < n; ++i) {
        int64_t r = x[i];
        if (o == 1) r = x[i] - x[i - 1];
        else if (o == 2) r = x[i] - 2*x[i - 1] + x[i - 2];
        else if (o == 3) r = x[i] - 3*x[i - 1] + 3*x[i - 2] - x[i - 3];
        else if (o == 4) r = x[i] - 4*x[i - 1] + 6*x[i - 2] - 4*x[i - 3] + x[i - 4];
        res[i - o] = r;
      }
      int po = 3;
      while (po > 0 && (n % (1 << po) != 0 || (n >> po) <= o)) --po;
      // per partition the best 4-bit Rice parameter (0..14)
      uint64_t total = 6 + 2 + 4;
      std::vector<int> ks(1 << po);
      int pos = 0;
      for (int part = 0; part < (1 << po); ++part) {
        const int count = po == 0 ? n - o : (n >> po) - (part == 0 ? o : 0);
        int bk = 0; uint64_t bb = ~0ull;
        for (int k = 0; k <= 14; ++k) {
          const uint64_t b = rice_bits(res.data() + pos, count, k);
          if (b < bb) { bb = b; bk = k; }
        }
        ks[part] = bk; total += 4 + bb; pos += count;
      }
      if (total + 16ull*o >= 16ull*n) {                    // prediction does not pay: VERBATIM
        bw.write(0, 1); bw.write(1, 6); bw.write(0, 1);
        for (int i = 0; i < n; ++i) bw.write((uint64_t)x[i], 16);
      } else {
        bw.write(0, 1); bw.write(8 + o, 6); bw.write(0, 1);
        for (int i = 0; i < o; ++i) bw.write((uint64_t)x[i], 16);
        bw.write(0, 2); bw.write(po, 4);
        pos = 0;
        for (int part = 0; part < (1 << po); ++part) {
          const int count = po == 0 ? n - o : (n >> po) - (part == 0 ? o : 0);
          const int k = ks[part];
          bw.write(k, 4);
          for (int i = 0; i < count; ++i) {
            const uint32_t u = zigzag(res[pos + i]);
            bw.unary(u >> k);
            bw.write(u & ((1u << k) - 1), k);
          }
          pos += count;
        }
      }
    }
    bw.align();
    const uint16_t c16 = crc16(buf.data() + f0, buf.size() - f0);
    buf.push_back((uint8_t)(c16 >> 8)); buf.push_back((uint8_t)(c16 & 0xff));
  }
  if (out == nullptr) return (int64_t)buf.size();          // size query
  if ((int64_t)buf.size() > capacity) return -2;
  memcpy(out, buf.data(), buf.size());
  return (int64_t)buf.size();
}
