// TFRecord container: per record
//   uint64 length (LE) | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data)
// Writer, sequential reader (with CRC verification) and an offset indexer used
// for per-rank sharding and shuffled random access.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "crc32c.h"

namespace {

struct Writer {
  FILE* f = nullptr;
};

struct Reader {
  FILE* f = nullptr;
  std::vector<uint8_t> buf;
  int verify = 1;
};

}  // namespace

extern "C" {

void* dtfa_tfrecord_writer_open(const char* path, int append) {
  FILE* f = fopen(path, append ? "ab" : "wb");
  if (!f) return nullptr;
  Writer* w = new Writer();
  w->f = f;
  return w;
}

int dtfa_tfrecord_write(void* h, const void* data, uint64_t n) {
  Writer* w = (Writer*)h;
  uint8_t head[12];
  memcpy(head, &n, 8);
  uint32_t c = dtfa::crc32c_mask(dtfa::crc32c(head, 8));
  memcpy(head + 8, &c, 4);
  uint32_t dc = dtfa::crc32c_mask(dtfa::crc32c(data, n));
  if (fwrite(head, 1, 12, w->f) != 12) return -1;
  if (n && fwrite(data, 1, n, w->f) != n) return -1;
  if (fwrite(&dc, 1, 4, w->f) != 4) return -1;
  return 0;
}

int dtfa_tfrecord_writer_flush(void* h) { return fflush(((Writer*)h)->f); }

int dtfa_tfrecord_writer_close(void* h) {
  Writer* w = (Writer*)h;
  int rc = fclose(w->f);
  delete w;
  return rc;
}

void* dtfa_tfrecord_reader_open(const char* path, int verify_crc) {
  FILE* f = fopen(path, "rb");
  if (!f) return nullptr;
  Reader* r = new Reader();
  r->f = f;
  r->verify = verify_crc;
  return r;
}

int dtfa_tfrecord_reader_seek(void* h, uint64_t offset) { return fseeko(((Reader*)h)->f, (off_t)offset, SEEK_SET); }

// Returns 1 and sets (*data, *n) on success (pointer valid until the next
// call), 0 at clean end of file, negative on corruption:
//   -1 truncated header, -2 bad length crc, -3 truncated data, -4 bad data crc
int dtfa_tfrecord_read_next(void* h, const uint8_t** data, uint64_t* n) {
  Reader* r = (Reader*)h;
  uint8_t head[12];
  size_t got = fread(head, 1, 12, r->f);
  if (got == 0) return 0;
  if (got != 12) return -1;
  uint64_t len;
  uint32_t c;
  memcpy(&len, head, 8);
  memcpy(&c, head + 8, 4);
  if (r->verify && dtfa::crc32c_mask(dtfa::crc32c(head, 8)) != c) return -2;
  if (len > (1ull << 40)) return -2;
  r->buf.resize(len + 4);
  if (fread(r->buf.data(), 1, len + 4, r->f) != len + 4) return -3;
  uint32_t dc;
  memcpy(&dc, r->buf.data() + len, 4);
  if (r->verify && dtfa::crc32c_mask(dtfa::crc32c(r->buf.data(), len)) != dc) return -4;
  *data = r->buf.data();
  *n = len;
  return 1;
}

int dtfa_tfrecord_reader_close(void* h) {
  Reader* r = (Reader*)h;
  int rc = fclose(r->f);
  delete r;
  return rc;
}

// Scans the file and stores the byte offset of every record header in
// offsets[0..cap) and the payload length in lengths[]; returns the record count
// (which may exceed cap: call again with a larger buffer) or negative on error.
long long dtfa_tfrecord_index(const char* path, uint64_t* offsets, uint64_t* lengths, long long cap) {
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  long long n = 0;
  uint64_t pos = 0;
  for (;;) {
    uint8_t head[12];
    size_t got = fread(head, 1, 12, f);
    if (got == 0) break;
    if (got != 12) { fclose(f); return -2; }
    uint64_t len;
    uint32_t c;
    memcpy(&len, head, 8);
    memcpy(&c, head + 8, 4);
    if (dtfa::crc32c_mask(dtfa::crc32c(head, 8)) != c) { fclose(f); return -3; }
    if (n < cap) {
      offsets[n] = pos;
      lengths[n] = len;
    }
    ++n;
    if (fseeko(f, (off_t)(len + 4), SEEK_CUR) != 0) { fclose(f); return -4; }
    pos += 12 + len + 4;
  }
  fclose(f);
  return n;
}

}  // extern "C"
