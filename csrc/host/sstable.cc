#include "sstable.h"

#include <string.h>

#include "crc32c.h"
#include "pb.h"

namespace dtfa {

static const uint64_t kMagic = 0xdb4775248b80fb57ull;

static void PutFixed32(std::string& s, uint32_t v) { s.append((const char*)&v, 4); }

void TableBuilder::BlockAdd(Block& b, const std::string& key, const std::string& value, int restart_interval) {
  size_t shared = 0;
  if (b.counter < restart_interval) {
    size_t m = std::min(b.last_key.size(), key.size());
    while (shared < m && b.last_key[shared] == key[shared]) ++shared;
  } else {
    b.restarts.push_back((uint32_t)b.buf.size());
    b.counter = 0;
  }
  pb::put_varint(b.buf, shared);
  pb::put_varint(b.buf, key.size() - shared);
  pb::put_varint(b.buf, value.size());
  b.buf.append(key.data() + shared, key.size() - shared);
  b.buf.append(value);
  b.last_key = key;
  b.counter++;
}

std::string TableBuilder::BlockFinish(Block& b) {
  std::string out = b.buf;
  for (uint32_t r : b.restarts) PutFixed32(out, r);
  PutFixed32(out, (uint32_t)b.restarts.size());
  b = Block();
  return out;
}

std::string TableBuilder::Handle(uint64_t off, uint64_t size) {
  std::string h;
  pb::put_varint(h, off);
  pb::put_varint(h, size);
  return h;
}

uint64_t TableBuilder::WriteRawBlock(const std::string& contents, uint64_t* size_out) {
  uint64_t off = file_.size();
  file_.append(contents);
  char type = 0;
  uint32_t crc = crc32c_extend(crc32c(contents.data(), contents.size()), &type, 1);
  file_.push_back(type);
  PutFixed32(file_, crc32c_mask(crc));
  *size_out = contents.size();
  return off;
}

void TableBuilder::FlushData() {
  if (data_.empty()) return;
  std::string last = data_.last_key;
  std::string contents = BlockFinish(data_);
  uint64_t size;
  uint64_t off = WriteRawBlock(contents, &size);
  pending_index_ = true;
  pending_key_ = last;
  pending_handle_ = Handle(off, size);
}

bool TableBuilder::Add(const std::string& key, const std::string& value) {
  if (have_key_ && !(last_key_ < key)) return false;
  if (pending_index_) {
    BlockAdd(index_, pending_key_, pending_handle_, 1);
    pending_index_ = false;
  }
  BlockAdd(data_, key, value, restart_interval_);
  last_key_ = key;
  have_key_ = true;
  if (data_.buf.size() + data_.restarts.size() * 4 + 4 >= block_size_) FlushData();
  return true;
}

std::string TableBuilder::Finish() {
  FlushData();
  if (pending_index_) {
    BlockAdd(index_, pending_key_, pending_handle_, 1);
    pending_index_ = false;
  }
  Block meta;
  uint64_t msize, isize;
  uint64_t moff = WriteRawBlock(BlockFinish(meta), &msize);
  uint64_t ioff = WriteRawBlock(BlockFinish(index_), &isize);
  std::string footer = Handle(moff, msize) + Handle(ioff, isize);
  footer.resize(40, '\0');
  footer.append((const char*)&kMagic, 8);
  file_.append(footer);
  return file_;
}

// ---------------------------------------------------------------------------
static bool ReadBlock(const std::string& file, uint64_t off, uint64_t size, std::string* contents, std::string* err) {
  if (off + size + 5 > file.size()) { *err = "block handle out of range"; return false; }
  const char* p = file.data() + off;
  if (p[size] != 0) { *err = "compressed blocks are not supported"; return false; }
  uint32_t stored;
  memcpy(&stored, p + size + 1, 4);
  uint32_t crc = crc32c(p, size + 1);
  if (crc32c_unmask(stored) != crc) { *err = "block checksum mismatch"; return false; }
  contents->assign(p, size);
  return true;
}

static bool IterBlock(const std::string& b, std::vector<std::pair<std::string, std::string>>* out, std::string* err) {
  if (b.size() < 4) { *err = "short block"; return false; }
  uint32_t nrestarts;
  memcpy(&nrestarts, b.data() + b.size() - 4, 4);
  if ((uint64_t)nrestarts * 4 + 4 > b.size()) { *err = "bad restart count"; return false; }
  size_t limit = b.size() - 4 - (size_t)nrestarts * 4;
  pb::Reader r(b.data(), limit);
  std::string key;
  while (!r.done()) {
    uint64_t shared = r.varint(), unshared = r.varint(), vlen = r.varint();
    if (!r.ok || shared > key.size() || unshared + vlen > (uint64_t)(r.end - r.p)) { *err = "bad block entry"; return false; }
    key.resize(shared);
    key.append((const char*)r.p, unshared);
    r.p += unshared;
    out->emplace_back(key, std::string((const char*)r.p, vlen));
    r.p += vlen;
  }
  return true;
}

bool TableReadAll(const std::string& file, std::vector<std::pair<std::string, std::string>>* out, std::string* err) {
  if (file.size() < 48) { *err = "file too short for a table footer"; return false; }
  const char* f = file.data() + file.size() - 48;
  uint64_t magic;
  memcpy(&magic, f + 40, 8);
  if (magic != kMagic) { *err = "bad table magic"; return false; }
  pb::Reader fr(f, 40);
  fr.varint(); fr.varint();  // metaindex handle
  uint64_t ioff = fr.varint(), isize = fr.varint();
  if (!fr.ok) { *err = "bad footer"; return false; }
  std::string index;
  if (!ReadBlock(file, ioff, isize, &index, err)) return false;
  std::vector<std::pair<std::string, std::string>> handles;
  if (!IterBlock(index, &handles, err)) return false;
  for (auto& kv : handles) {
    pb::Reader hr(kv.second.data(), kv.second.size());
    uint64_t off = hr.varint(), size = hr.varint();
    if (!hr.ok) { *err = "bad block handle"; return false; }
    std::string block;
    if (!ReadBlock(file, off, size, &block, err)) return false;
    if (!IterBlock(block, out, err)) return false;
  }
  return true;
}

}  // namespace dtfa
