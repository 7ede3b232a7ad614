// Baseline (sequential, Huffman, 8-bit) JPEG entropy decoder.
//
// The decode is split between host and device: this file parses the markers and
// undoes the Huffman / run-length coding, producing the quantised DCT
// coefficients of every 8x8 block (natural row-major order inside a block,
// blocks in raster order per component, planes padded to whole MCUs).  The
// GPU kernel in csrc/hip/jpeg.hip then dequantises, runs the 8x8 inverse DCT,
// upsamples chroma and converts YCbCr to RGB.  Progressive / arithmetic /
// 12-bit files are reported as unsupported so the caller can fall back to a
// software decoder.
#include <stdint.h>
#include <string.h>

#include <vector>

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool present = false;
  uint8_t vals[256];
  int mincode[17], maxcode[18], valptr[17];
  // 9-bit fast path: (length << 8) | value, 0 = not resolvable in 9 bits
  uint16_t fast[512];
  // AC tables: code AND coefficient bits inside the 9-bit window: (coefficient << 8) | (run << 4) | total bits, 0 = no
  int16_t fast_ac[512];
  void build_fast_ac() {
    for (int i = 0; i < 512; ++i) {
      fast_ac[i] = 0;
      const int f = fast[i];
      if (!f) continue;
      const int rs = f & 0xff, len = f >> 8;
      const int run = rs >> 4, mag = rs & 15;
      if (mag == 0 || len + mag > 9) continue;
      int k = ((i << len) & 511) >> (9 - mag);   // the mag bits behind the code
      if (k < (1 << (mag - 1))) k += (int)(~0u << mag) + 1;
      if (k >= -128 && k <= 127) fast_ac[i] = (int16_t)(k * 256 + run * 16 + len + mag);
    }
  }
  // False for a code-length histogram that is not a prefix code (more codes of some length than that length has
  // left: Kraft sum > 1).  Such a table would index ``fast`` past its end.
  bool build(const uint8_t* bits, const uint8_t* v, int nvals) {
    present = false;
    if (nvals < 0 || nvals > 256) return false;
    memcpy(vals, v, nvals);
    int code = 0, k = 0;
    memset(fast, 0, sizeof(fast));
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      if (code + (int)bits[l] > (1 << l) || k + (int)bits[l] > nvals) return false;
      for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
        if (l <= 9) {
          int lo = code << (9 - l), hi = lo + (1 << (9 - l));
          for (int c = lo; c < hi; ++c) fast[c] = (uint16_t)((l << 8) | vals[k]);
        }
      }
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    build_fast_ac();
    present = true;
    return true;
  }
};

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t buf = 0;
  int cnt = 0;
  bool hit_marker = false;
  int pad_bytes = 0;  // zero bytes fed after the end of the entropy-coded data
  void fill() {
    // common case: the next eight bytes hold no 0xff (no stuffing, no marker): append as many whole bytes as fit at once
    if (!hit_marker && p + 8 <= end && cnt <= 56) {
      uint64_t w;
      memcpy(&w, p, 8);
      w = __builtin_bswap64(w);
      const uint64_t x = ~w;   // a 0xff byte of w is a zero byte of x
      if (!((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull)) {
        const int nbytes = (64 - cnt) >> 3;
        const int nbits = nbytes * 8;
        uint64_t add = w >> cnt;                              // first new byte right behind the cnt valid bits
        if (cnt + nbits < 64) add &= ~0ull << (64 - cnt - nbits);   // drop the partial byte behind the whole ones
        buf |= add;
        cnt += nbits;
        p += nbytes;
        return;
      }
    }
    while (cnt <= 56) {
      uint8_t b = 0;
      if (hit_marker || p >= end) ++pad_bytes;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xff) {
          if (p + 1 < end && p[1] == 0x00) {
            p += 2;
          } else {
            hit_marker = true;  // leave p at the marker; feed zeros
            b = 0;
          }
        } else {
          ++p;
        }
      }
      buf |= (uint64_t)b << (56 - cnt);
      cnt += 8;
    }
  }
  inline int peek(int n) { return (int)(buf >> (64 - n)); }
  inline void skip(int n) { buf <<= n; cnt -= n; }
  inline int get(int n) {
    if (n == 0) return 0;
    if (cnt < n) fill();
    int v = peek(n);
    skip(n);
    return v;
  }
  void reset_at_restart() {
    buf = 0;
    cnt = 0;
    hit_marker = false;
    pad_bytes = 0;
  }
};

// value of an s-bit coefficient field (s >= 1): fields with a leading 0 bit are the negative half
inline int extend(int v, int s) { return v + (((v >> (s - 1)) - 1) & (int)((~0u << s) + 1)); }

inline int decode_symbol(BitReader& br, const Huff& h) {
  if (br.cnt < 16) br.fill();
  int f = h.fast[br.peek(9)];
  if (f) {
    br.skip(f >> 8);
    return f & 0xff;
  }
  int code = br.peek(9);
  int l = 9;
  br.skip(9);
  for (;;) {
    if (l > 16) return -1;
    if (code <= h.maxcode[l] && h.maxcode[l] >= 0 && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    code = (code << 1) | br.get(1);
    ++l;
  }
}

}  // namespace

extern "C" {

struct DtfaJpegInfo {
  int width, height, ncomp;
  int hs[4], vs[4];
  int blocks_w[4], blocks_h[4];   // per component, padded to whole MCUs
  long long coef_offset[4];       // start of the component's coefficients, in int16 units
  long long total_coefs;          // total int16 coefficients
  unsigned short qt[4][64];       // quantisation tables, natural order, per component
  int hmax, vmax;
  int restart_interval;
  int sof;                        // SOF marker type (0 = baseline, 1 = extended sequential, 2 = progressive, ...)
};

// Largest image accepted (pixels).  The header alone sizes the coefficient buffer, so an unchecked 65535 x 65535
// header would make a server allocate ~25 GB of pinned memory for a few bytes of input.
const long long kMaxPixels = 1ll << 26;

// error codes: -1 malformed, -2 unsupported coding (progressive / arithmetic / 12-bit / >4 components),
// -3 coefficient buffer too small, -4 image larger than kMaxPixels
static int parse(const uint8_t* d, size_t n, DtfaJpegInfo* info, Huff dc[4], Huff ac[4], int comp_dc[4], int comp_ac[4],
                 const uint8_t** scan, uint16_t qtab[4][64], int comp_tq[4], bool want_scan) {
  if (n < 4 || d[0] != 0xff || d[1] != 0xd8) return -1;
  size_t p = 2;
  memset(info, 0, sizeof(*info));
  info->sof = -1;
  int comp_id[4] = {0, 0, 0, 0};
  bool have_q[4] = {false, false, false, false};
  while (p + 4 <= n) {
    if (d[p] != 0xff) return -1;
    while (p < n && d[p] == 0xff) ++p;  // fill bytes
    if (p >= n) return -1;
    uint8_t m = d[p++];
    if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;
    if (m == 0xd9) return -1;
    if (p + 2 > n) return -1;
    size_t len = ((size_t)d[p] << 8) | d[p + 1];
    if (len < 2 || p + len > n) return -1;
    const uint8_t* s = d + p + 2;
    size_t sl = len - 2;
    if (m == 0xdb) {  // DQT
      size_t q = 0;
      while (q < sl) {
        int pq = s[q] >> 4, tq = s[q] & 15;
        ++q;
        if (tq > 3) return -1;
        if (q + (pq ? 128 : 64) > sl) return -1;
        for (int i = 0; i < 64; ++i) {
          uint16_t v = pq ? (uint16_t)((s[q] << 8) | s[q + 1]) : s[q];
          q += pq ? 2 : 1;
          qtab[tq][kZigzag[i]] = v;
        }
        have_q[tq] = true;
      }
    } else if (m == 0xc0 || m == 0xc1 || m == 0xc2 || (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) {
      info->sof = m - 0xc0;
      if (m != 0xc0 && m != 0xc1) return -2;
      if (sl < 6) return -1;
      if (s[0] != 8) return -2;
      info->height = (s[1] << 8) | s[2];
      info->width = (s[3] << 8) | s[4];
      info->ncomp = s[5];
      if (info->ncomp < 1 || info->ncomp > 4 || sl < 6 + 3 * (size_t)info->ncomp) return -2;
      if (info->width <= 0 || info->height <= 0) return -1;
      if ((long long)info->width * info->height > kMaxPixels) return -4;
      for (int c = 0; c < info->ncomp; ++c) {
        comp_id[c] = s[6 + 3 * c];
        info->hs[c] = s[7 + 3 * c] >> 4;
        info->vs[c] = s[7 + 3 * c] & 15;
        comp_tq[c] = s[8 + 3 * c] & 3;
        if (info->hs[c] < 1 || info->hs[c] > 4 || info->vs[c] < 1 || info->vs[c] > 4) return -1;
        if (info->hs[c] > info->hmax) info->hmax = info->hs[c];
        if (info->vs[c] > info->vmax) info->vmax = info->vs[c];
      }
    } else if (m == 0xc4) {  // DHT
      size_t q = 0;
      while (q < sl) {
        if (q + 17 > sl) return -1;
        int tc = s[q] >> 4, th = s[q] & 15;
        if (tc > 1 || th > 3) return -1;
        uint8_t bits[17];
        bits[0] = 0;
        int total = 0;
        for (int i = 1; i <= 16; ++i) {
          bits[i] = s[q + i];
          total += bits[i];
        }
        q += 17;
        if (total > 256 || q + total > sl) return -1;
        if (!(tc ? ac : dc)[th].build(bits, s + q, total)) return -1;
        q += total;
      }
    } else if (m == 0xdd) {  // DRI
      if (sl < 2) return -1;
      info->restart_interval = (s[0] << 8) | s[1];
    } else if (m == 0xda) {  // SOS
      if (info->sof < 0) return -1;
      if (sl < 1) return -1;
      int ns = s[0];
      if (ns != info->ncomp || sl < 1 + 2 * (size_t)ns + 3) return -2;  // non-interleaved scans unsupported
      for (int i = 0; i < ns; ++i) {
        int cid = s[1 + 2 * i], tbl = s[2 + 2 * i];
        int c = -1;
        for (int k = 0; k < info->ncomp; ++k)
          if (comp_id[k] == cid) c = k;
        if (c < 0) return -1;
        comp_dc[c] = tbl >> 4;
        comp_ac[c] = tbl & 15;
        if (comp_dc[c] > 3 || comp_ac[c] > 3) return -1;
      }
      // geometry
      int mcux = (info->width + 8 * info->hmax - 1) / (8 * info->hmax);
      int mcuy = (info->height + 8 * info->vmax - 1) / (8 * info->vmax);
      long long off = 0;
      for (int c = 0; c < info->ncomp; ++c) {
        info->blocks_w[c] = mcux * info->hs[c];
        info->blocks_h[c] = mcuy * info->vs[c];
        info->coef_offset[c] = off;
        off += (long long)info->blocks_w[c] * info->blocks_h[c] * 64;
        if (!have_q[comp_tq[c]]) return -1;
        memcpy(info->qt[c], qtab[comp_tq[c]], 128);
      }
      info->total_coefs = off;
      *scan = d + p + len;
      return 0;
    }
    p += len;
    (void)want_scan;
  }
  return -1;
}

int dtfa_jpeg_parse_header(const uint8_t* data, uint64_t n, DtfaJpegInfo* info) {
  Huff dc[4], ac[4];
  int cdc[4] = {0}, cac[4] = {0}, ctq[4] = {0};
  uint16_t qtab[4][64];
  memset(qtab, 0, sizeof(qtab));
  const uint8_t* scan = nullptr;
  return parse(data, (size_t)n, info, dc, ac, cdc, cac, &scan, qtab, ctq, false);
}

// coefs must hold info.total_coefs int16 values; it is zero-filled here.
int dtfa_jpeg_decode_coefs(const uint8_t* data, uint64_t n, DtfaJpegInfo* info, int16_t* coefs, long long cap) {
  Huff dc[4], ac[4];
  int cdc[4] = {0}, cac[4] = {0}, ctq[4] = {0};
  uint16_t qtab[4][64];
  memset(qtab, 0, sizeof(qtab));
  const uint8_t* scan = nullptr;
  int rc = parse(data, (size_t)n, info, dc, ac, cdc, cac, &scan, qtab, ctq, true);
  if (rc != 0) return rc;
  if (cap < info->total_coefs) return -3;
  for (int c = 0; c < info->ncomp; ++c)
    if (!dc[cdc[c]].present || !ac[cac[c]].present) return -1;
  memset(coefs, 0, (size_t)info->total_coefs * 2);
  BitReader br;
  br.p = scan;
  br.end = data + n;
  int pred[4] = {0, 0, 0, 0};
  int mcux = info->blocks_w[0] / info->hs[0];
  int mcuy = info->blocks_h[0] / info->vs[0];
  int restart = info->restart_interval;
  int count = 0;
  for (int my = 0; my < mcuy; ++my) {
    for (int mx = 0; mx < mcux; ++mx) {
      if (restart && count == restart) {
        // byte-align, consume RSTn
        br.reset_at_restart();
        const uint8_t* q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) ++q;
        if (q + 1 >= br.end) return -1;
        br.p = q + 2;
        pred[0] = pred[1] = pred[2] = pred[3] = 0;
        count = 0;
      }
      for (int c = 0; c < info->ncomp; ++c) {
        const Huff& hd = dc[cdc[c]];
        const Huff& ha = ac[cac[c]];
        for (int by = 0; by < info->vs[c]; ++by) {
          for (int bx = 0; bx < info->hs[c]; ++bx) {
            long long bi = (long long)(my * info->vs[c] + by) * info->blocks_w[c] + (mx * info->hs[c] + bx);
            int16_t* blk = coefs + info->coef_offset[c] + bi * 64;
            int s = decode_symbol(br, hd);
            if (s < 0 || s > 11) return -1;
            int diff = s ? extend(br.get(s), s) : 0;
            pred[c] += diff;
            blk[0] = (int16_t)pred[c];
            for (int k = 1; k < 64;) {
              // 32 bits in the window cover the longest code (16) and the longest coefficient (15) without a refill
              if (br.cnt < 32) br.fill();
              const int w9 = br.peek(9);
              const int fa = ha.fast_ac[w9];
              if (fa) {   // run, size and the coefficient itself from one lookup
                k += (fa >> 4) & 15;
                if (k > 63) return -1;
                br.skip(fa & 15);
                blk[kZigzag[k]] = (int16_t)(fa >> 8);
                ++k;
                continue;
              }
              int rs;
              const int f = ha.fast[w9];
              if (f) {
                br.skip(f >> 8);
                rs = f & 0xff;
              } else {
                rs = decode_symbol(br, ha);
                if (rs < 0) return -1;
              }
              const int r = rs >> 4, sz = rs & 15;
              if (sz == 0) {
                if (r == 15) { k += 16; continue; }
                break;
              }
              k += r;
              if (k > 63) return -1;
              if (br.cnt < sz) br.fill();
              const int v = br.peek(sz);
              br.skip(sz);
              blk[kZigzag[k]] = (int16_t)extend(v, sz);
              ++k;
            }
          }
        }
      }
      ++count;
      // the reader feeds zeros past the end of the data; a few bytes of look-ahead are normal, an MCU decoded
      // entirely from padding means the scan is truncated
      if (br.pad_bytes > 32) return -1;
    }
  }
  return 0;
}

int dtfa_jpeg_info_size() { return (int)sizeof(DtfaJpegInfo); }

}  // extern "C"
