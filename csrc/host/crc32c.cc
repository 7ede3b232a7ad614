// CRC32-C (Castagnoli).  Uses the SSE4.2 crc32 instruction when the CPU has
// it, otherwise a slicing-by-1 table.
#include "crc32c.h"

#include <nmmintrin.h>

namespace dtfa {

static uint32_t g_table[256];
static bool g_table_init = false;
static void init_table() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82f63b78u : c >> 1;
    g_table[i] = c;
  }
  g_table_init = true;
}

static uint32_t crc_sw(uint32_t crc, const uint8_t* p, size_t n) {
  if (!g_table_init) init_table();
  uint32_t c = crc ^ 0xffffffffu;
  for (size_t i = 0; i < n; ++i) c = g_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
  return c ^ 0xffffffffu;
}

__attribute__((target("sse4.2"))) static uint32_t crc_hw(uint32_t crc, const uint8_t* p, size_t n) {
  uint64_t c = crc ^ 0xffffffffu;
  while (n && ((uintptr_t)p & 7)) {
    c = _mm_crc32_u8((uint32_t)c, *p++);
    --n;
  }
  while (n >= 8) {
    c = _mm_crc32_u64(c, *(const uint64_t*)p);
    p += 8;
    n -= 8;
  }
  while (n--) c = _mm_crc32_u8((uint32_t)c, *p++);
  return (uint32_t)c ^ 0xffffffffu;
}

uint32_t crc32c_extend(uint32_t crc, const void* data, size_t n) {
  static int has_hw = -1;
  if (has_hw < 0) has_hw = __builtin_cpu_supports("sse4.2") ? 1 : 0;
  return has_hw ? crc_hw(crc, (const uint8_t*)data, n) : crc_sw(crc, (const uint8_t*)data, n);
}

}  // namespace dtfa

extern "C" uint32_t dtfa_crc32c(const void* data, size_t n) { return dtfa::crc32c(data, n); }
extern "C" uint32_t dtfa_crc32c_extend(uint32_t crc, const void* data, size_t n) {
  return dtfa::crc32c_extend(crc, data, n);
}
extern "C" uint32_t dtfa_crc32c_masked(const void* data, size_t n) { return dtfa::crc32c_mask(dtfa::crc32c(data, n)); }
extern "C" uint32_t dtfa_crc32c_sw_for_test(const void* data, size_t n) {
  return dtfa::crc_sw(0, (const uint8_t*)data, n);
}
