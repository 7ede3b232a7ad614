// TensorFlow Saver-V2 "tensor bundle" checkpoints:
//   <prefix>.index                 sorted table: "" -> BundleHeaderProto, name -> BundleEntryProto
//   <prefix>.data-00000-of-00001   raw little-endian tensor bytes, in key order
//
//   BundleHeaderProto { int32 num_shards = 1; Endianness endianness = 2; VersionDef version = 3 { int32 producer = 1; } }
//   BundleEntryProto  { DataType dtype = 1; TensorShapeProto shape = 2 { repeated Dim dim = 2 { int64 size = 1; } }
//                       int32 shard_id = 3; int64 offset = 4; int64 size = 5; fixed32 crc32c = 6; }
// The writer streams tensor bytes to a temporary data file as they are added
// (keys must arrive sorted) and renames data + index into place at the end, so a
// reader never observes a half-written checkpoint.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "crc32c.h"
#include "pb.h"
#include "sstable.h"

namespace {

struct Entry {
  std::string name;
  int dtype = 0;
  std::vector<int64_t> dims;
  int shard = 0;
  uint64_t offset = 0, size = 0;
  uint32_t crc = 0;  // masked
};

struct BundleWriter {
  std::string prefix, data_tmp, data_final;
  FILE* f = nullptr;
  uint64_t pos = 0;
  std::vector<Entry> entries;
  std::string last;
};

struct BundleReader {
  std::vector<Entry> entries;
  int num_shards = 1;
  std::string error;
};

std::string encode_entry(const Entry& e) {
  std::string s;
  pb::put_varint_field(s, 1, (uint64_t)e.dtype);
  std::string shape;
  for (int64_t d : e.dims) {
    std::string dim;
    pb::put_varint_field(dim, 1, (uint64_t)d);
    pb::put_string_field(shape, 2, dim);
  }
  pb::put_string_field(s, 2, shape);
  if (e.shard) pb::put_varint_field(s, 3, (uint64_t)e.shard);
  if (e.offset) pb::put_varint_field(s, 4, e.offset);
  if (e.size) pb::put_varint_field(s, 5, e.size);
  pb::put_fixed32_field(s, 6, e.crc);
  return s;
}

bool decode_entry(const std::string& v, Entry* e) {
  pb::Reader r(v.data(), v.size());
  int field, wire;
  while (r.tag(field, wire)) {
    if (field == 1 && wire == 0) e->dtype = (int)r.varint();
    else if (field == 2 && wire == 2) {
      const uint8_t* p; size_t n;
      if (!r.bytes(p, n)) return false;
      pb::Reader sr(p, n);
      int sf, sw;
      while (sr.tag(sf, sw)) {
        if (sf == 2 && sw == 2) {
          const uint8_t* dp; size_t dn;
          if (!sr.bytes(dp, dn)) return false;
          pb::Reader dr(dp, dn);
          int df, dw;
          int64_t size = 0;
          while (dr.tag(df, dw)) {
            if (df == 1 && dw == 0) size = (int64_t)dr.varint();
            else dr.skip(dw);
          }
          e->dims.push_back(size);
        } else sr.skip(sw);
      }
      if (!sr.ok) return false;
    } else if (field == 3 && wire == 0) e->shard = (int)r.varint();
    else if (field == 4 && wire == 0) e->offset = r.varint();
    else if (field == 5 && wire == 0) e->size = r.varint();
    else if (field == 6 && wire == 5) e->crc = r.fixed32();
    else r.skip(wire);
  }
  return r.ok;
}

bool read_file(const std::string& path, std::string* out) {
  std::ifstream in(path, std::ios::binary);
  if (!in) return false;
  std::stringstream ss;
  ss << in.rdbuf();
  *out = ss.str();
  return true;
}

}  // namespace

extern "C" {

void* dtfa_bundle_writer_new(const char* prefix) {
  BundleWriter* w = new BundleWriter();
  w->prefix = prefix;
  w->data_final = w->prefix + ".data-00000-of-00001";
  w->data_tmp = w->data_final + ".tmp" + std::to_string((long)getpid());
  w->f = fopen(w->data_tmp.c_str(), "wb");
  if (!w->f) { delete w; return nullptr; }
  return w;
}

// returns 0 on success, -1 if keys are not strictly increasing, -2 on IO error
int dtfa_bundle_add(void* h, const char* name, int dtype, const int64_t* dims, int ndim, const void* data,
                    uint64_t nbytes) {
  BundleWriter* w = (BundleWriter*)h;
  std::string key(name);
  if (key.empty() || (!w->entries.empty() && !(w->last < key))) return -1;
  Entry e;
  e.name = key;
  e.dtype = dtype;
  e.dims.assign(dims, dims + ndim);
  e.offset = w->pos;
  e.size = nbytes;
  e.crc = dtfa::crc32c_mask(dtfa::crc32c(data, nbytes));
  if (nbytes && fwrite(data, 1, nbytes, w->f) != nbytes) return -2;
  w->pos += nbytes;
  w->entries.push_back(std::move(e));
  w->last = key;
  return 0;
}

int dtfa_bundle_writer_finish(void* h) {
  BundleWriter* w = (BundleWriter*)h;
  int rc = 0;
  if (fclose(w->f) != 0) rc = -2;
  w->f = nullptr;
  dtfa::TableBuilder tb;
  std::string header;
  pb::put_varint_field(header, 1, 1);  // num_shards
  std::string version;
  pb::put_varint_field(version, 1, 1);  // producer
  pb::put_string_field(header, 3, version);
  tb.Add("", header);
  for (auto& e : w->entries) tb.Add(e.name, encode_entry(e));
  std::string table = tb.Finish();
  std::string idx_final = w->prefix + ".index";
  std::string idx_tmp = idx_final + ".tmp" + std::to_string((long)getpid());
  FILE* f = fopen(idx_tmp.c_str(), "wb");
  if (!f || fwrite(table.data(), 1, table.size(), f) != table.size()) rc = -2;
  if (f && fclose(f) != 0) rc = -2;
  if (rc == 0) {
    if (rename(w->data_tmp.c_str(), w->data_final.c_str()) != 0) rc = -3;
    if (rc == 0 && rename(idx_tmp.c_str(), idx_final.c_str()) != 0) rc = -3;
  }
  if (rc != 0) {
    remove(w->data_tmp.c_str());
    remove(idx_tmp.c_str());
  }
  delete w;
  return rc;
}

void dtfa_bundle_writer_abort(void* h) {
  BundleWriter* w = (BundleWriter*)h;
  if (w->f) fclose(w->f);
  remove(w->data_tmp.c_str());
  delete w;
}

void* dtfa_bundle_reader_open(const char* prefix) {
  BundleReader* r = new BundleReader();
  std::string file;
  if (!read_file(std::string(prefix) + ".index", &file)) {
    r->error = "cannot read index file";
    return r;
  }
  std::vector<std::pair<std::string, std::string>> kv;
  if (!dtfa::TableReadAll(file, &kv, &r->error)) return r;
  for (auto& p : kv) {
    if (p.first.empty()) {
      pb::Reader hr(p.second.data(), p.second.size());
      int field, wire;
      while (hr.tag(field, wire)) {
        if (field == 1 && wire == 0) r->num_shards = (int)hr.varint();
        else if (field == 2 && wire == 0) { if (hr.varint() != 0) r->error = "big-endian bundles are not supported"; }
        else hr.skip(wire);
      }
      continue;
    }
    Entry e;
    e.name = p.first;
    if (!decode_entry(p.second, &e)) { r->error = "bad bundle entry for " + p.first; break; }
    r->entries.push_back(std::move(e));
  }
  return r;
}

const char* dtfa_bundle_reader_error(void* h) {
  BundleReader* r = (BundleReader*)h;
  return r->error.empty() ? nullptr : r->error.c_str();
}
int dtfa_bundle_num_entries(void* h) { return (int)((BundleReader*)h)->entries.size(); }
int dtfa_bundle_num_shards(void* h) { return ((BundleReader*)h)->num_shards; }

// dims must have room for 8 entries; returns ndim or -1
int dtfa_bundle_entry(void* h, int i, const char** name, int* dtype, int64_t* dims, int* shard, uint64_t* offset,
                      uint64_t* size, uint32_t* masked_crc) {
  BundleReader* r = (BundleReader*)h;
  if (i < 0 || i >= (int)r->entries.size()) return -1;
  const Entry& e = r->entries[i];
  if (e.dims.size() > 8) return -1;
  *name = e.name.c_str();
  *dtype = e.dtype;
  for (size_t k = 0; k < e.dims.size(); ++k) dims[k] = e.dims[k];
  *shard = e.shard;
  *offset = e.offset;
  *size = e.size;
  *masked_crc = e.crc;
  return (int)e.dims.size();
}

void dtfa_bundle_reader_close(void* h) { delete (BundleReader*)h; }

// Generic table access (tests / tools): build a table file from parallel arrays.
int dtfa_table_write(const char* path, int n, const char** keys, const uint64_t* key_lens, const char** vals,
                     const uint64_t* val_lens, int block_size) {
  dtfa::TableBuilder tb(block_size > 0 ? (size_t)block_size : 262144);
  for (int i = 0; i < n; ++i)
    if (!tb.Add(std::string(keys[i], key_lens[i]), std::string(vals[i], val_lens[i]))) return -1;
  std::string t = tb.Finish();
  FILE* f = fopen(path, "wb");
  if (!f) return -2;
  size_t w = fwrite(t.data(), 1, t.size(), f);
  fclose(f);
  return w == t.size() ? 0 : -2;
}

// Returns a handle over all (key, value) pairs of a table file.
struct TableDump { std::vector<std::pair<std::string, std::string>> kv; std::string err; };
void* dtfa_table_read(const char* path) {
  TableDump* d = new TableDump();
  std::string file;
  if (!read_file(path, &file)) { d->err = "cannot read file"; return d; }
  dtfa::TableReadAll(file, &d->kv, &d->err);
  return d;
}
const char* dtfa_table_error(void* h) { TableDump* d = (TableDump*)h; return d->err.empty() ? nullptr : d->err.c_str(); }
int dtfa_table_count(void* h) { return (int)((TableDump*)h)->kv.size(); }
int dtfa_table_get(void* h, int i, const char** k, uint64_t* kn, const char** v, uint64_t* vn) {
  TableDump* d = (TableDump*)h;
  if (i < 0 || i >= (int)d->kv.size()) return -1;
  *k = d->kv[i].first.data(); *kn = d->kv[i].first.size();
  *v = d->kv[i].second.data(); *vn = d->kv[i].second.size();
  return 0;
}
void dtfa_table_free(void* h) { delete (TableDump*)h; }

}  // extern "C"
