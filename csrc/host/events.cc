// TensorBoard event-file writer.
//
// events.out.tfevents.<unix time>.<host> is a TFRecord stream of Event protos:
//   Event { double wall_time = 1; int64 step = 2; string file_version = 3; Summary summary = 5; }
//   Summary { repeated Value value = 1 { string tag = 1; float simple_value = 2; HistogramProto histo = 5; } }
//   HistogramProto { double min = 1, max = 2, num = 3, sum = 4, sum_squares = 5;
//                    repeated double bucket_limit = 6 [packed]; repeated double bucket = 7 [packed]; }
// The first record carries file_version "brain.Event:2".  Histograms use
// TensorFlow's default exponential bucket boundaries (1e-12 * 1.1^k, mirrored
// for negatives) and collapse runs of empty buckets the way TensorFlow does.
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "pb.h"

extern "C" void* dtfa_tfrecord_writer_open(const char* path, int append);
extern "C" int dtfa_tfrecord_write(void* h, const void* data, uint64_t n);
extern "C" int dtfa_tfrecord_writer_flush(void* h);
extern "C" int dtfa_tfrecord_writer_close(void* h);

namespace {

const std::vector<double>& default_limits() {
  static std::vector<double> limits;
  if (limits.empty()) {
    std::vector<double> pos;
    double v = 1e-12;
    while (v < 1e20) {
      pos.push_back(v);
      v *= 1.1;
    }
    pos.push_back(DBL_MAX);
    for (size_t i = pos.size(); i-- > 0;) limits.push_back(-pos[i]);
    limits.push_back(0.0);
    limits.insert(limits.end(), pos.begin(), pos.end());
  }
  return limits;
}

struct EventWriter {
  void* rec = nullptr;
  std::string path;
  std::string summary;  // values of the event being assembled
  int64_t step = 0;
  double wall = 0;
};

double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  return (double)ts.tv_sec + ts.tv_nsec * 1e-9;
}

void write_event(EventWriter* w, double wall, int64_t step, const std::string& extra) {
  std::string ev;
  pb::put_double_field(ev, 1, wall);
  if (step) pb::put_varint_field(ev, 2, (uint64_t)step);
  ev.append(extra);
  dtfa_tfrecord_write(w->rec, ev.data(), ev.size());
}

}  // namespace

extern "C" {

void* dtfa_events_open(const char* logdir, const char* suffix) {
  char host[256] = "localhost";
  gethostname(host, sizeof(host) - 1);
  EventWriter* w = new EventWriter();
  char name[1024];
  snprintf(name, sizeof(name), "%s/events.out.tfevents.%010ld.%s%s", logdir, (long)time(nullptr), host,
           suffix ? suffix : "");
  w->path = name;
  w->rec = dtfa_tfrecord_writer_open(name, 0);
  if (!w->rec) { delete w; return nullptr; }
  std::string fv;
  pb::put_string_field(fv, 3, "brain.Event:2");
  write_event(w, now_s(), 0, fv);
  dtfa_tfrecord_writer_flush(w->rec);
  return w;
}

const char* dtfa_events_path(void* h) { return ((EventWriter*)h)->path.c_str(); }

void dtfa_events_begin(void* h, int64_t step, double wall_time) {
  EventWriter* w = (EventWriter*)h;
  w->summary.clear();
  w->step = step;
  w->wall = wall_time > 0 ? wall_time : now_s();
}

void dtfa_events_add_scalar(void* h, const char* tag, float value) {
  EventWriter* w = (EventWriter*)h;
  std::string val;
  pb::put_string_field(val, 1, tag);
  pb::put_float_field(val, 2, value);
  pb::put_string_field(w->summary, 1, val);
}

void dtfa_events_add_histogram(void* h, const char* tag, const float* values, int64_t n) {
  EventWriter* w = (EventWriter*)h;
  const std::vector<double>& limits = default_limits();
  std::vector<double> buckets(limits.size(), 0.0);
  double mn = DBL_MAX, mx = -DBL_MAX, sum = 0, sq = 0;
  for (int64_t i = 0; i < n; ++i) {
    double v = values[i];
    size_t b = std::upper_bound(limits.begin(), limits.end(), v) - limits.begin();
    if (b >= limits.size()) b = limits.size() - 1;
    buckets[b] += 1.0;
    mn = std::min(mn, v);
    mx = std::max(mx, v);
    sum += v;
    sq += v * v;
  }
  if (n == 0) mn = mx = 0;
  std::string lim_packed, cnt_packed;
  for (size_t i = 0; i < buckets.size();) {
    double end = limits[i], count = buckets[i];
    ++i;
    if (count <= 0.0) {
      while (i < buckets.size() && buckets[i] <= 0.0) {
        end = limits[i];
        count = buckets[i];
        ++i;
      }
    }
    lim_packed.append((const char*)&end, 8);
    cnt_packed.append((const char*)&count, 8);
  }
  std::string histo;
  pb::put_double_field(histo, 1, mn);
  pb::put_double_field(histo, 2, mx);
  pb::put_double_field(histo, 3, (double)n);
  pb::put_double_field(histo, 4, sum);
  pb::put_double_field(histo, 5, sq);
  pb::put_string_field(histo, 6, lim_packed);
  pb::put_string_field(histo, 7, cnt_packed);
  std::string val;
  pb::put_string_field(val, 1, tag);
  pb::put_string_field(val, 5, histo);
  pb::put_string_field(w->summary, 1, val);
}

int dtfa_events_end(void* h) {
  EventWriter* w = (EventWriter*)h;
  std::string extra;
  pb::put_string_field(extra, 5, w->summary);
  write_event(w, w->wall, w->step, extra);
  w->summary.clear();
  return 0;
}

int dtfa_events_flush(void* h) { return dtfa_tfrecord_writer_flush(((EventWriter*)h)->rec); }

int dtfa_events_close(void* h) {
  EventWriter* w = (EventWriter*)h;
  int rc = dtfa_tfrecord_writer_close(w->rec);
  delete w;
  return rc;
}

}  // extern "C"
