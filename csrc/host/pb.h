// Minimal protobuf wire-format helpers (encode + decode) for the handful of
// TensorFlow messages the framework reads and writes: tf.train.Example,
// BundleHeaderProto / BundleEntryProto, Event / Summary / HistogramProto.
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

namespace pb {

inline void put_varint(std::string& s, uint64_t v) {
  while (v >= 0x80) {
    s.push_back((char)((v & 0x7f) | 0x80));
    v >>= 7;
  }
  s.push_back((char)v);
}
inline void put_tag(std::string& s, int field, int wire) { put_varint(s, ((uint64_t)field << 3) | (uint64_t)wire); }
inline void put_varint_field(std::string& s, int field, uint64_t v) {
  put_tag(s, field, 0);
  put_varint(s, v);
}
inline void put_bytes_field(std::string& s, int field, const void* p, size_t n) {
  put_tag(s, field, 2);
  put_varint(s, n);
  s.append((const char*)p, n);
}
inline void put_string_field(std::string& s, int field, const std::string& v) {
  put_bytes_field(s, field, v.data(), v.size());
}
inline void put_fixed32_field(std::string& s, int field, uint32_t v) {
  put_tag(s, field, 5);
  s.append((const char*)&v, 4);
}
inline void put_float_field(std::string& s, int field, float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  put_fixed32_field(s, field, u);
}
inline void put_double_field(std::string& s, int field, double v) {
  put_tag(s, field, 1);
  s.append((const char*)&v, 8);
}

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  Reader(const void* data, size_t n) : p((const uint8_t*)data), end((const uint8_t*)data + n) {}
  bool done() const { return p >= end || !ok; }
  uint64_t varint() {
    uint64_t v = 0;
    int shift = 0;
    while (p < end && shift < 64) {
      uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
      shift += 7;
    }
    ok = false;
    return 0;
  }
  // reads a tag; returns false at end of buffer
  bool tag(int& field, int& wire) {
    if (p >= end) return false;
    uint64_t t = varint();
    if (!ok) return false;
    field = (int)(t >> 3);
    wire = (int)(t & 7);
    return true;
  }
  bool bytes(const uint8_t*& q, size_t& n) {
    uint64_t len = varint();
    if (!ok || len > (uint64_t)(end - p)) {
      ok = false;
      return false;
    }
    q = p;
    n = (size_t)len;
    p += len;
    return true;
  }
  uint32_t fixed32() {
    if (end - p < 4) { ok = false; return 0; }
    uint32_t v;
    memcpy(&v, p, 4);
    p += 4;
    return v;
  }
  uint64_t fixed64() {
    if (end - p < 8) { ok = false; return 0; }
    uint64_t v;
    memcpy(&v, p, 8);
    p += 8;
    return v;
  }
  void skip(int wire) {
    switch (wire) {
      case 0: varint(); break;
      case 1: fixed64(); break;
      case 2: { const uint8_t* q; size_t n; bytes(q, n); break; }
      case 5: fixed32(); break;
      default: ok = false;
    }
  }
};

}  // namespace pb
