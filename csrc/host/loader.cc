// Threaded TFRecord batch loader (the native counterpart of the tf.data
// pipeline the reference builds: parallel reads, parallel map, repeat, batch,
// shuffle, prefetch).
//
// * the file is indexed once (record offsets), so every epoch is an exact
//   example-level permutation and ranks read disjoint shards
//   (indices rank, rank+world, ...);
// * `threads` workers each assemble whole batches: read the records, parse
//   the tf.Example (image_raw / label / height / width / channel) and copy the
//   raw pixel bytes back-to-back into a caller-owned (pinned) slot buffer;
// * `slots` batches are prefetched ahead of the consumer; batches are delivered
//   in order.
// The consumer uploads a slot with one async copy and lets the GPU resize /
// normalise kernel read the variable-size images through the offset table.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

extern "C" {
long long dtfa_tfrecord_index(const char* path, uint64_t* offsets, uint64_t* lengths, long long cap);
void* dtfa_example_parse(const void* data, uint64_t n);
void dtfa_example_free(void* h);
int dtfa_example_get_bytes(void* h, const char* key, int idx, const uint8_t** data, uint64_t* n);
int dtfa_example_get_int64(void* h, const char* key, int64_t* out, int cap);
}

namespace {

enum SlotState { FREE = 0, FILLING = 1, READY = 2, IN_USE = 3 };

struct Slot {
  uint8_t* buf = nullptr;
  uint64_t cap = 0;
  int state = FREE;
  long long batch_index = -1;
  int count = 0;
  int error = 0;
  std::vector<uint64_t> offsets;
  std::vector<int32_t> heights, widths, channels, labels;
};

struct Loader {
  std::string path;
  std::vector<uint64_t> rec_off, rec_len;
  int rank = 0, world = 1, batch = 32, epochs = 1;
  bool shuffle = true, drop_remainder = false;
  uint64_t seed = 0;
  std::vector<Slot> slots;
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv;
  bool stop = false;
  long long next_to_fill = 0;     // next batch index a worker will claim
  long long next_to_deliver = 0;  // next batch index the consumer gets
  long long total_batches = 0;
  long long per_epoch = 0;        // examples this rank sees per epoch
  std::vector<std::vector<uint32_t>> epoch_perm_cache;  // small cache keyed by epoch % 2
  std::vector<long long> epoch_perm_id;
  std::mutex perm_mu;

  const std::vector<uint32_t>& perm_for(long long epoch, std::vector<uint32_t>& scratch) {
    // per-rank shard of a seeded permutation of all records
    scratch.clear();
    std::vector<uint32_t> all(rec_off.size());
    for (size_t i = 0; i < all.size(); ++i) all[i] = (uint32_t)i;
    if (shuffle) {
      std::mt19937_64 rng(seed * 0x9e3779b97f4a7c15ull + (uint64_t)epoch);
      std::shuffle(all.begin(), all.end(), rng);
    }
    for (size_t i = (size_t)rank; i < all.size(); i += (size_t)world) scratch.push_back(all[i]);
    return scratch;
  }

  void worker() {
    FILE* f = fopen(path.c_str(), "rb");
    std::vector<uint8_t> rec;
    std::vector<uint32_t> perm;
    long long perm_epoch = -1;
    for (;;) {
      long long bi;
      Slot* s;
      {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
          if (stop || next_to_fill >= total_batches) { if (f) fclose(f); return; }
          bi = next_to_fill;
          s = &slots[(size_t)(bi % (long long)slots.size())];
          if (s->state == FREE) break;
          cv.wait(lk);
        }
        ++next_to_fill;
        s->state = FILLING;
        s->batch_index = bi;
      }
      // examples of this batch: global example positions [bi*batch, ...) within the rank's stream
      int err = f ? 0 : -10;
      uint64_t pos = 0;
      int count = 0;
      s->offsets.clear(); s->heights.clear(); s->widths.clear(); s->channels.clear(); s->labels.clear();
      long long first = bi * (long long)batch;
      long long stream_total = per_epoch * (long long)epochs;
      for (int k = 0; k < batch && err == 0; ++k) {
        long long g = first + k;
        if (g >= stream_total) break;
        long long epoch = g / per_epoch;
        long long within = g % per_epoch;
        if (epoch != perm_epoch) { perm_for(epoch, perm); perm_epoch = epoch; }
        uint32_t ri = perm[(size_t)within];
        uint64_t len = rec_len[ri];
        rec.resize(len);
        if (fseeko(f, (off_t)(rec_off[ri] + 12), SEEK_SET) != 0 || fread(rec.data(), 1, len, f) != len) { err = -11; break; }
        void* ex = dtfa_example_parse(rec.data(), len);
        if (!ex) { err = -12; break; }
        const uint8_t* img; uint64_t in;
        int64_t h = 0, w = 0, c = 0, lab = 0;
        if (dtfa_example_get_bytes(ex, "image_raw", 0, &img, &in) != 0 || dtfa_example_get_int64(ex, "height", &h, 1) < 1 ||
            dtfa_example_get_int64(ex, "width", &w, 1) < 1 || dtfa_example_get_int64(ex, "channel", &c, 1) < 1 ||
            dtfa_example_get_int64(ex, "label", &lab, 1) < 1) {
          err = -13;
        } else if ((uint64_t)(h * w * c) != in) {
          err = -14;
        } else if (pos + in > s->cap) {
          err = -15;
        } else {
          memcpy(s->buf + pos, img, in);
          s->offsets.push_back(pos);
          s->heights.push_back((int32_t)h); s->widths.push_back((int32_t)w); s->channels.push_back((int32_t)c);
          s->labels.push_back((int32_t)lab);
          pos += (in + 15) & ~15ull;  // keep every image 16-byte aligned
          ++count;
        }
        dtfa_example_free(ex);
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        s->count = count;
        s->error = err;
        s->state = READY;
      }
      cv.notify_all();
    }
  }
};

}  // namespace

extern "C" {

struct DtfaBatch {
  int slot;
  int count;
  const uint64_t* offsets;
  const int32_t* heights;
  const int32_t* widths;
  const int32_t* channels;
  const int32_t* labels;
  uint64_t bytes_used;
};

// slot_bufs: `slots` caller-owned buffers of `slot_cap` bytes each (ideally pinned).
void* dtfa_loader_open(const char* path, int rank, int world, int batch, int threads, int slots, uint8_t** slot_bufs,
                       uint64_t slot_cap, uint64_t seed, int shuffle, int epochs, int drop_remainder) {
  if (world < 1 || rank < 0 || rank >= world || batch < 1 || threads < 1 || slots < 1) return nullptr;
  long long n = dtfa_tfrecord_index(path, nullptr, nullptr, 0);
  if (n <= 0) return nullptr;
  Loader* L = new Loader();
  L->path = path;
  L->rec_off.resize((size_t)n);
  L->rec_len.resize((size_t)n);
  if (dtfa_tfrecord_index(path, L->rec_off.data(), L->rec_len.data(), n) != n) { delete L; return nullptr; }
  L->rank = rank; L->world = world; L->batch = batch; L->epochs = epochs; L->shuffle = shuffle != 0;
  L->seed = seed; L->drop_remainder = drop_remainder != 0;
  L->per_epoch = (n - rank + world - 1) / world;
  if (L->per_epoch <= 0) { delete L; return nullptr; }
  long long stream = L->per_epoch * (long long)epochs;
  L->total_batches = drop_remainder ? stream / batch : (stream + batch - 1) / batch;
  L->slots.resize((size_t)slots);
  for (int i = 0; i < slots; ++i) { L->slots[i].buf = slot_bufs[i]; L->slots[i].cap = slot_cap; }
  for (int t = 0; t < threads; ++t) L->workers.emplace_back([L] { L->worker(); });
  return L;
}

long long dtfa_loader_num_records(void* h) { return (long long)((Loader*)h)->rec_off.size(); }
long long dtfa_loader_num_batches(void* h) { return ((Loader*)h)->total_batches; }

// 1 = batch delivered, 0 = end of data, negative = error while assembling the batch
int dtfa_loader_next(void* h, DtfaBatch* out) {
  Loader* L = (Loader*)h;
  std::unique_lock<std::mutex> lk(L->mu);
  if (L->next_to_deliver >= L->total_batches) return 0;
  Slot& s = L->slots[(size_t)(L->next_to_deliver % (long long)L->slots.size())];
  L->cv.wait(lk, [&] { return L->stop || (s.state == READY && s.batch_index == L->next_to_deliver); });
  if (L->stop) return 0;
  s.state = IN_USE;
  ++L->next_to_deliver;
  if (s.error) return s.error;
  out->slot = (int)(&s - L->slots.data());
  out->count = s.count;
  out->offsets = s.offsets.data();
  out->heights = s.heights.data();
  out->widths = s.widths.data();
  out->channels = s.channels.data();
  out->labels = s.labels.data();
  out->bytes_used = s.count ? s.offsets.back() + (uint64_t)s.heights.back() * s.widths.back() * s.channels.back() : 0;
  return 1;
}

void dtfa_loader_release(void* h, int slot) {
  Loader* L = (Loader*)h;
  {
    std::lock_guard<std::mutex> lk(L->mu);
    if (slot >= 0 && slot < (int)L->slots.size()) L->slots[(size_t)slot].state = FREE;
  }
  L->cv.notify_all();
}

void dtfa_loader_close(void* h) {
  Loader* L = (Loader*)h;
  {
    std::lock_guard<std::mutex> lk(L->mu);
    L->stop = true;
  }
  L->cv.notify_all();
  for (auto& t : L->workers) t.join();
  delete L;
}

}  // extern "C"
