// LevelDB-format immutable sorted table (the container of TensorFlow's
// checkpoint ".index" files): prefix-compressed key blocks with restart points,
// a 5-byte trailer per block (compression type 0 + masked crc32c), an index
// block, an empty metaindex block and a 48-byte footer ending in the magic
// 0xdb4775248b80fb57.
#pragma once
#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

namespace dtfa {

class TableBuilder {
 public:
  explicit TableBuilder(size_t block_size = 262144, int restart_interval = 16)
      : block_size_(block_size), restart_interval_(restart_interval) {}
  // keys must be added in strictly increasing bytewise order
  bool Add(const std::string& key, const std::string& value);
  // returns the complete file contents
  std::string Finish();

 private:
  struct Block {
    std::string buf;
    std::vector<uint32_t> restarts{0};
    int counter = 0;
    std::string last_key;
    bool empty() const { return buf.empty(); }
  };
  static void BlockAdd(Block& b, const std::string& key, const std::string& value, int restart_interval);
  static std::string BlockFinish(Block& b);
  void FlushData();
  static std::string Handle(uint64_t off, uint64_t size);
  uint64_t WriteRawBlock(const std::string& contents, uint64_t* size_out);

  size_t block_size_;
  int restart_interval_;
  Block data_;
  Block index_;
  std::string file_;
  std::string last_key_;
  bool have_key_ = false;
  bool pending_index_ = false;
  std::string pending_handle_;
  std::string pending_key_;
};

// Parses a whole table; returns false on any format / checksum error.
bool TableReadAll(const std::string& file, std::vector<std::pair<std::string, std::string>>* out, std::string* err);

}  // namespace dtfa
