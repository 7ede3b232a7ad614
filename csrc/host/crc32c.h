#pragma once
#include <stddef.h>
#include <stdint.h>
namespace dtfa {
uint32_t crc32c_extend(uint32_t crc, const void* data, size_t n);
inline uint32_t crc32c(const void* data, size_t n) { return crc32c_extend(0, data, n); }
// TensorFlow / LevelDB "masked" CRC stored next to the data it covers.
inline uint32_t crc32c_mask(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }
inline uint32_t crc32c_unmask(uint32_t m) {
  uint32_t rot = m - 0xa282ead8u;
  return (rot >> 17) | (rot << 15);
}
}  // namespace dtfa
