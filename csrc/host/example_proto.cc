// tf.train.Example codec.
//
//   Example  { Features features = 1; }
//   Features { map<string, Feature> feature = 1; }   // map entry: key = 1, value = 2
//   Feature  { oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3; } }
//   BytesList { repeated bytes value = 1; }
//   FloatList { repeated float value = 1 [packed]; }
//   Int64List { repeated int64 value = 1 [packed]; }
//
// The builder emits map entries in key order (what a sorted std::map gives
// TensorFlow's C++ serializer in deterministic mode).  The parser accepts
// packed and unpacked repeated scalars.
#include <stdint.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "pb.h"

namespace {

struct Feature {
  int kind = 0;  // 1 bytes, 2 float, 3 int64
  std::vector<std::string> bytes;
  std::vector<float> floats;
  std::vector<int64_t> ints;
};

struct Builder {
  std::map<std::string, Feature> feats;
  std::string out;
};

struct Parsed {
  // views into the caller's buffer for bytes values (zero copy)
  struct F {
    int kind = 0;
    std::vector<std::pair<const uint8_t*, size_t>> bytes;
    std::vector<float> floats;
    std::vector<int64_t> ints;
  };
  std::map<std::string, F> feats;
};

bool parse_feature(const uint8_t* p, size_t n, Parsed::F& f) {
  pb::Reader r(p, n);
  int field, wire;
  while (r.tag(field, wire)) {
    if (wire != 2) { r.skip(wire); continue; }
    const uint8_t* q; size_t m;
    if (!r.bytes(q, m)) return false;
    pb::Reader lr(q, m);
    int lf, lw;
    if (field == 1) {
      f.kind = 1;
      while (lr.tag(lf, lw)) {
        if (lf == 1 && lw == 2) {
          const uint8_t* b; size_t bn;
          if (!lr.bytes(b, bn)) return false;
          f.bytes.emplace_back(b, bn);
        } else lr.skip(lw);
      }
    } else if (field == 2) {
      f.kind = 2;
      while (lr.tag(lf, lw)) {
        if (lf == 1 && lw == 2) {
          const uint8_t* b; size_t bn;
          if (!lr.bytes(b, bn)) return false;
          for (size_t i = 0; i + 4 <= bn; i += 4) { float v; memcpy(&v, b + i, 4); f.floats.push_back(v); }
        } else if (lf == 1 && lw == 5) {
          uint32_t u = lr.fixed32(); float v; memcpy(&v, &u, 4); f.floats.push_back(v);
        } else lr.skip(lw);
      }
    } else if (field == 3) {
      f.kind = 3;
      while (lr.tag(lf, lw)) {
        if (lf == 1 && lw == 2) {
          const uint8_t* b; size_t bn;
          if (!lr.bytes(b, bn)) return false;
          pb::Reader vr(b, bn);
          while (!vr.done()) f.ints.push_back((int64_t)vr.varint());
          if (!vr.ok) return false;
        } else if (lf == 1 && lw == 0) {
          f.ints.push_back((int64_t)lr.varint());
        } else lr.skip(lw);
      }
    }
    if (!lr.ok) return false;
  }
  return r.ok;
}

}  // namespace

extern "C" {

void* dtfa_example_builder_new() { return new Builder(); }
void dtfa_example_builder_free(void* h) { delete (Builder*)h; }

void dtfa_example_add_bytes(void* h, const char* key, const void* data, uint64_t n) {
  Feature& f = ((Builder*)h)->feats[key];
  f.kind = 1;
  f.bytes.emplace_back((const char*)data, (size_t)n);
}
void dtfa_example_add_int64(void* h, const char* key, const int64_t* v, int n) {
  Feature& f = ((Builder*)h)->feats[key];
  f.kind = 3;
  f.ints.insert(f.ints.end(), v, v + n);
}
void dtfa_example_add_float(void* h, const char* key, const float* v, int n) {
  Feature& f = ((Builder*)h)->feats[key];
  f.kind = 2;
  f.floats.insert(f.floats.end(), v, v + n);
}

int dtfa_example_serialize(void* h, const uint8_t** data, uint64_t* n) {
  Builder* b = (Builder*)h;
  std::string features;
  for (auto& kv : b->feats) {
    const Feature& f = kv.second;
    std::string list, feat;
    if (f.kind == 1) {
      for (auto& s : f.bytes) pb::put_string_field(list, 1, s);
      pb::put_string_field(feat, 1, list);
    } else if (f.kind == 2) {
      std::string packed((const char*)f.floats.data(), f.floats.size() * 4);
      if (!f.floats.empty()) pb::put_string_field(list, 1, packed);
      pb::put_string_field(feat, 2, list);
    } else {
      std::string packed;
      for (int64_t v : f.ints) pb::put_varint(packed, (uint64_t)v);
      if (!f.ints.empty()) pb::put_string_field(list, 1, packed);
      pb::put_string_field(feat, 3, list);
    }
    std::string entry;
    pb::put_string_field(entry, 1, kv.first);
    pb::put_string_field(entry, 2, feat);
    pb::put_string_field(features, 1, entry);
  }
  b->out.clear();
  pb::put_string_field(b->out, 1, features);
  *data = (const uint8_t*)b->out.data();
  *n = b->out.size();
  return 0;
}

// The parsed object keeps pointers into `data`; the caller must keep that
// buffer alive while it uses the handle.
void* dtfa_example_parse(const void* data, uint64_t n) {
  Parsed* ex = new Parsed();
  pb::Reader r(data, n);
  int field, wire;
  while (r.tag(field, wire)) {
    if (field != 1 || wire != 2) { r.skip(wire); continue; }
    const uint8_t* fp; size_t fn;
    if (!r.bytes(fp, fn)) break;
    pb::Reader fr(fp, fn);
    int ff, fw;
    while (fr.tag(ff, fw)) {
      if (ff != 1 || fw != 2) { fr.skip(fw); continue; }
      const uint8_t* ep; size_t en;
      if (!fr.bytes(ep, en)) break;
      pb::Reader er(ep, en);
      int ef, ew;
      std::string key;
      const uint8_t* vp = nullptr; size_t vn = 0;
      while (er.tag(ef, ew)) {
        if (ew != 2) { er.skip(ew); continue; }
        const uint8_t* q; size_t m;
        if (!er.bytes(q, m)) break;
        if (ef == 1) key.assign((const char*)q, m);
        else if (ef == 2) { vp = q; vn = m; }
      }
      if (!er.ok) { fr.ok = false; break; }
      Parsed::F f;
      if (vp && !parse_feature(vp, vn, f)) { fr.ok = false; break; }
      ex->feats[key] = std::move(f);
    }
    if (!fr.ok) { r.ok = false; break; }
  }
  if (!r.ok) { delete ex; return nullptr; }
  return ex;
}

void dtfa_example_free(void* h) { delete (Parsed*)h; }

int dtfa_example_num_features(void* h) { return (int)((Parsed*)h)->feats.size(); }

// kind of a feature: 0 missing, 1 bytes, 2 float, 3 int64; *count = number of values
int dtfa_example_feature_info(void* h, const char* key, int* count) {
  Parsed* ex = (Parsed*)h;
  auto it = ex->feats.find(key);
  if (it == ex->feats.end()) { *count = 0; return 0; }
  const Parsed::F& f = it->second;
  *count = (int)(f.kind == 1 ? f.bytes.size() : f.kind == 2 ? f.floats.size() : f.ints.size());
  return f.kind;
}

int dtfa_example_get_bytes(void* h, const char* key, int idx, const uint8_t** data, uint64_t* n) {
  Parsed* ex = (Parsed*)h;
  auto it = ex->feats.find(key);
  if (it == ex->feats.end() || it->second.kind != 1 || idx < 0 || idx >= (int)it->second.bytes.size()) return -1;
  *data = it->second.bytes[idx].first;
  *n = it->second.bytes[idx].second;
  return 0;
}

int dtfa_example_get_int64(void* h, const char* key, int64_t* out, int cap) {
  Parsed* ex = (Parsed*)h;
  auto it = ex->feats.find(key);
  if (it == ex->feats.end() || it->second.kind != 3) return -1;
  int n = (int)it->second.ints.size();
  for (int i = 0; i < n && i < cap; ++i) out[i] = it->second.ints[i];
  return n;
}

int dtfa_example_get_float(void* h, const char* key, float* out, int cap) {
  Parsed* ex = (Parsed*)h;
  auto it = ex->feats.find(key);
  if (it == ex->feats.end() || it->second.kind != 2) return -1;
  int n = (int)it->second.floats.size();
  for (int i = 0; i < n && i < cap; ++i) out[i] = it->second.floats[i];
  return n;
}

// name of the i-th feature in key order (for listing)
const char* dtfa_example_feature_name(void* h, int i) {
  Parsed* ex = (Parsed*)h;
  for (auto& kv : ex->feats)
    if (i-- == 0) return kv.first.c_str();
  return nullptr;
}

}  // extern "C"
