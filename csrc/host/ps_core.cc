// Data plane of a TCP parameter-server task.
//
// The reference's ps task is TensorFlow's gRPC server: it holds the variables, answers variable reads and runs
// ApplyAdam for every gradient a worker sends (recognition_flower/distributed/train_dataset.py:141-145,162-165,
// 188-191).  parallel/ps.py speaks a small framed protocol over TCP instead; this file is the part of it that runs
// once per worker step and moves megabytes: PULL (variables to the worker) and PUSH (gradient from the worker,
// optimizer step on the ps).  Everything else -- initialisation, checkpoints, status, shutdown -- stays in Python,
// which hands a connection to dtfa_ps_serve() and gets it back when a frame that is not PULL / PUSH arrives.
//
//   * one thread per connection (the Python thread that accepted it, inside dtfa_ps_serve with the GIL released);
//   * a PULL is answered straight out of the parameter buffer (header + payload in one sendmsg, no copy, no lock:
//     like TensorFlow's ps, reads race with concurrent applies at the granularity of single floats);
//   * a PUSH is received into the connection's own buffer and applied shard by shard, each shard under its own
//     mutex, so gradients of different workers are applied concurrently on different parts of the variables; the
//     loops are plain multiply-adds over contiguous floats that the compiler vectorises (AVX2 where the CPU has it),
//     and the shards of a large gradient are dealt to a small pool of helper threads shared by all connections
//     (sleeping between gradients: no spinning beside the connection threads);
//   * bounded staleness, worker clocks, the global step and the debug assertions of parallel/ps.py are kept here,
//     behind one control mutex that is never held while bytes move.
//
// Wire format (little endian): 32-byte header  magic "DTPS" | u32 op | i32 worker | i32 status | i64 clock/step |
// u64 payload bytes, then the payload.
#include <arpa/inet.h>
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/uio.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t OP_PULL = 4, OP_PUSH = 5;
constexpr int ST_OK = 0, ST_NOT_INIT = 1, ST_ERROR = 2, ST_STOPPED = 3;
constexpr int NSHARD = 16;

#pragma pack(push, 1)
struct Header {
  char magic[4];
  uint32_t op;
  int32_t worker;
  int32_t status;
  int64_t clock;
  uint64_t nbytes;
};
#pragma pack(pop)
static_assert(sizeof(Header) == 32, "frame header");

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct PsCore {
  // control state
  std::mutex mu;
  std::condition_variable cv;
  int staleness = -1;
  double worker_timeout = 30.0;
  bool debug = false;
  bool initialized = false;
  std::atomic<bool> stop{false};
  std::map<int, long long> clocks;
  std::map<int, double> last_seen;
  long long max_lag_seen = 0;
  long long clock_advances = 0;
  std::vector<std::string> violations;
  long long global_step = 0;   // guarded by mu
  long long apply_count = 0;   // guarded by mu
  // variables and optimizer (buffers owned by the caller)
  float* w = nullptr;
  float* s0 = nullptr;
  float* s1 = nullptr;
  size_t n = 0;
  int kind = 0;  // 0 Adam, 1 momentum
  double lr0 = 1e-3, decay_rate = 1.0;
  long long decay_steps = 0;
  double b1 = 0.9, b2 = 0.999;   // (kept in double: 1 - b2 formed in float is off by 1e-5 relative)
  float eps = 1e-8f, mom = 0.9f, wd = 0.f;
  int nesterov = 0;
  int apply_threads = 4;
  std::mutex shard_mu[NSHARD];
  // helper pool: shard-sized pieces of the gradients being applied
  struct Task {
    const float* g;
    int shard;
    float lr_t;
    std::atomic<int>* remaining;
  };
  std::mutex qmu;
  std::condition_variable qcv, done_cv;
  std::deque<Task> tasks;
  std::vector<std::thread> helpers;
  bool pool_stop = false;
  std::atomic<unsigned> rotate{0};
};

bool recv_all(int fd, void* buf, size_t n) {
  char* p = (char*)buf;
  while (n) {
    ssize_t k = ::recv(fd, p, n, 0);
    if (k == 0) return false;
    if (k < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += k;
    n -= (size_t)k;
  }
  return true;
}

bool send_frame(int fd, uint32_t op, int worker, int status, long long clock, const void* payload, size_t nbytes) {
  Header h;
  memcpy(h.magic, "DTPS", 4);
  h.op = op;
  h.worker = worker;
  h.status = status;
  h.clock = clock;
  h.nbytes = nbytes;
  iovec iov[2] = {{&h, sizeof(h)}, {(void*)payload, nbytes}};
  size_t total = sizeof(h) + nbytes, sent = 0;
  int cnt = nbytes ? 2 : 1;
  iovec* cur = iov;
  while (sent < total) {
    msghdr m;
    memset(&m, 0, sizeof(m));
    m.msg_iov = cur;
    m.msg_iovlen = cnt;
    ssize_t k = ::sendmsg(fd, &m, MSG_NOSIGNAL);
    if (k < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    sent += (size_t)k;
    size_t left = (size_t)k;
    while (left && cnt) {
      if (left >= cur->iov_len) {
        left -= cur->iov_len;
        ++cur;
        --cnt;
      } else {
        cur->iov_base = (char*)cur->iov_base + left;
        cur->iov_len -= left;
        left = 0;
      }
    }
  }
  return true;
}

long long live_min_clock(PsCore* c, double now, bool* any) {
  long long m = 0;
  *any = false;
  for (auto& kv : c->clocks) {
    auto it = c->last_seen.find(kv.first);
    const double seen = it == c->last_seen.end() ? 0.0 : it->second;
    if (now - seen <= c->worker_timeout) {
      if (!*any || kv.second < m) m = kv.second;
      *any = true;
    }
  }
  return m;
}

// Hold a pull back while the worker is more than `staleness` steps ahead of the slowest live worker.
void wait_for_staleness(PsCore* c, int worker, long long clock) {
  if (c->staleness < 0) return;
  std::unique_lock<std::mutex> lk(c->mu);
  while (!c->stop.load()) {
    const double now = now_s();
    c->last_seen[worker] = now;
    bool any;
    const long long m = live_min_clock(c, now, &any);
    if (!any || clock - m <= c->staleness) {
      if (any && clock - m > c->max_lag_seen) c->max_lag_seen = clock - m;
      if (c->debug && c->max_lag_seen > c->staleness)
        c->violations.push_back("pull of worker " + std::to_string(worker) + " admitted at lag " + std::to_string(clock - m) +
                                " > " + std::to_string(c->staleness));
      return;
    }
    c->cv.wait_for(lk, std::chrono::milliseconds(50));
  }
}

// TensorFlow-1 ApplyAdam with the bias correction folded into lr_t; momentum SGD with optional Nesterov and L2.
__attribute__((target_clones("avx2", "default"))) void adam_range(float* __restrict__ w, float* __restrict__ m,
                                                                  float* __restrict__ v, const float* __restrict__ g,
                                                                  size_t n, float lr_t, float b1, float b2, float c1,
                                                                  float c2, float eps) {
  for (size_t i = 0; i < n; ++i) {
    const float gi = g[i];
    const float mi = b1 * m[i] + c1 * gi;
    const float vi = b2 * v[i] + c2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    w[i] = w[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

__attribute__((target_clones("avx2", "default"))) void momentum_range(float* __restrict__ w, float* __restrict__ buf,
                                                                      const float* __restrict__ g, size_t n, float lr,
                                                                      float mu, float wd, int nesterov) {
  for (size_t i = 0; i < n; ++i) {
    const float gp = g[i] + wd * w[i];
    const float bi = mu * buf[i] + gp;
    buf[i] = bi;
    w[i] = w[i] - lr * (nesterov ? gp + mu * bi : bi);
  }
}

double rate_at(const PsCore* c, long long step) {
  if (c->decay_steps <= 0) return c->lr0;
  return c->lr0 * pow(c->decay_rate, (double)step / (double)c->decay_steps);
}

void run_task(PsCore* c, const PsCore::Task& t) {
  const size_t per = (c->n + NSHARD - 1) / NSHARD;
  const size_t lo = (size_t)t.shard * per;
  if (lo < c->n) {
    const size_t cnt = (lo + per <= c->n ? per : c->n - lo);
    std::lock_guard<std::mutex> lk(c->shard_mu[t.shard]);
    if (c->kind == 0)
      adam_range(c->w + lo, c->s0 + lo, c->s1 + lo, t.g + lo, cnt, t.lr_t, (float)c->b1, (float)c->b2, (float)(1.0 - c->b1),
                 (float)(1.0 - c->b2), c->eps);
    else
      momentum_range(c->w + lo, c->s0 + lo, t.g + lo, cnt, t.lr_t, c->mom, c->wd, c->nesterov);
  }
  if (t.remaining->fetch_sub(1) == 1) {
    std::lock_guard<std::mutex> lk(c->qmu);
    c->done_cv.notify_all();
  }
}

void helper_loop(PsCore* c) {
  for (;;) {
    PsCore::Task t;
    {
      std::unique_lock<std::mutex> lk(c->qmu);
      c->qcv.wait(lk, [c] { return c->pool_stop || !c->tasks.empty(); });
      if (c->tasks.empty()) return;   // (stopping)
      t = c->tasks.front();
      c->tasks.pop_front();
    }
    run_task(c, t);
  }
}

void apply_gradient(PsCore* c, const float* g, long long step) {
  const double lr = rate_at(c, step);
  float lr_t = (float)lr;
  if (c->kind == 0) {
    const double t = (double)(step + 1);
    lr_t = (float)(lr * sqrt(1.0 - pow(c->b2, t)) / (1.0 - pow(c->b1, t)));
  }
  std::atomic<int> remaining{NSHARD};
  // concurrent gradients start at different shards, so that they meet on a shard mutex as late as possible
  const unsigned first = c->rotate.fetch_add(5u) % NSHARD;
  const bool pooled = c->n >= (1u << 18) && !c->helpers.empty();
  if (!pooled) {
    for (int k = 0; k < NSHARD; ++k) run_task(c, PsCore::Task{g, (int)((first + k) % NSHARD), lr_t, &remaining});
    return;
  }
  {
    std::lock_guard<std::mutex> lk(c->qmu);
    for (int k = 0; k < NSHARD; ++k) c->tasks.push_back(PsCore::Task{g, (int)((first + k) % NSHARD), lr_t, &remaining});
  }
  c->qcv.notify_all();
  // the connection thread works on the queue too (its own pieces or another gradient's), then waits for its last ones
  for (;;) {
    PsCore::Task t;
    {
      std::unique_lock<std::mutex> lk(c->qmu);
      if (remaining.load() == 0) return;
      if (c->tasks.empty()) {
        c->done_cv.wait(lk, [&] { return remaining.load() == 0 || !c->tasks.empty(); });
        if (remaining.load() == 0) return;
        continue;
      }
      t = c->tasks.front();
      c->tasks.pop_front();
    }
    run_task(c, t);
  }
}

}  // namespace

extern "C" {

void* dtfa_ps_new(int staleness, double worker_timeout, int debug) {
  PsCore* c = new PsCore();
  c->staleness = staleness;
  c->worker_timeout = worker_timeout;
  c->debug = debug != 0;
  return c;
}

void dtfa_ps_free(void* core) {
  PsCore* c = (PsCore*)core;
  {
    std::lock_guard<std::mutex> lk(c->qmu);
    c->pool_stop = true;
  }
  c->qcv.notify_all();
  for (auto& t : c->helpers) t.join();
  delete c;
}

// The variables (w), the optimizer slots (s0; s1 for Adam) and the optimizer's constants.  The buffers stay owned by
// the caller and must outlive the core.  Marks the task initialised.
void dtfa_ps_set_data(void* core, float* w, float* s0, float* s1, long long n, int kind, double lr0, long long decay_steps,
                      double decay_rate, double b1, double b2, float eps, float mom, float wd, int nesterov,
                      long long global_step, int apply_threads) {
  PsCore* c = (PsCore*)core;
  std::lock_guard<std::mutex> lk(c->mu);
  c->w = w;
  c->s0 = s0;
  c->s1 = s1;
  c->n = (size_t)n;
  c->kind = kind;
  c->lr0 = lr0;
  c->decay_steps = decay_steps;
  c->decay_rate = decay_rate;
  c->b1 = b1;
  c->b2 = b2;
  c->eps = eps;
  c->mom = mom;
  c->wd = wd;
  c->nesterov = nesterov;
  c->global_step = global_step;
  c->apply_threads = apply_threads > 0 ? apply_threads : 1;
  while ((int)c->helpers.size() < c->apply_threads - 1) c->helpers.emplace_back(helper_loop, c);
  c->initialized = true;
  c->cv.notify_all();
}

void dtfa_ps_set_step(void* core, long long step) {
  PsCore* c = (PsCore*)core;
  std::lock_guard<std::mutex> lk(c->mu);
  c->global_step = step;
}

void dtfa_ps_hello(void* core, int worker, long long clock) {
  PsCore* c = (PsCore*)core;
  std::lock_guard<std::mutex> lk(c->mu);
  c->last_seen[worker] = now_s();
  c->clocks.emplace(worker, clock);
}

void dtfa_ps_bye(void* core, int worker) {
  PsCore* c = (PsCore*)core;
  std::lock_guard<std::mutex> lk(c->mu);
  c->clocks.erase(worker);
  c->last_seen.erase(worker);
  c->cv.notify_all();
}

void dtfa_ps_stop(void* core) {
  PsCore* c = (PsCore*)core;
  c->stop.store(true);
  std::lock_guard<std::mutex> lk(c->mu);
  c->cv.notify_all();
}

// Exclusive access to the variables and slots (checkpoint reads, restores): every shard mutex, in order.
void dtfa_ps_lock(void* core) {
  PsCore* c = (PsCore*)core;
  for (int s = 0; s < NSHARD; ++s) c->shard_mu[s].lock();
}
void dtfa_ps_unlock(void* core) {
  PsCore* c = (PsCore*)core;
  for (int s = NSHARD - 1; s >= 0; --s) c->shard_mu[s].unlock();
}

// out[0] global step, out[1] apply count, out[2] max lag seen, out[3] number of violations, out[4] number of clocks;
// clocks: up to `cap` (worker, clock) pairs.
void dtfa_ps_status(void* core, long long* out, long long* clocks, int cap) {
  PsCore* c = (PsCore*)core;
  std::lock_guard<std::mutex> lk(c->mu);
  out[0] = c->global_step;
  out[1] = c->apply_count;
  out[2] = c->max_lag_seen;
  out[3] = (long long)c->violations.size();
  int k = 0;
  for (auto& kv : c->clocks) {
    if (k >= cap) break;
    clocks[2 * k] = kv.first;
    clocks[2 * k + 1] = kv.second;
    ++k;
  }
  out[4] = k;
}

// Violation i as text into buf (returns its length, 0 when there is none).
int dtfa_ps_violation(void* core, int i, char* buf, int cap) {
  PsCore* c = (PsCore*)core;
  std::lock_guard<std::mutex> lk(c->mu);
  if (i < 0 || i >= (int)c->violations.size() || cap <= 0) return 0;
  const std::string& s = c->violations[i];
  const int n = (int)s.size() < cap - 1 ? (int)s.size() : cap - 1;
  memcpy(buf, s.data(), n);
  buf[n] = 0;
  return n;
}

// Serve PULL / PUSH frames on the connected socket fd until a frame of another kind arrives (its header is copied to
// hdr_out and 1 is returned: the caller handles it, payload included, and calls again), the peer closes (0) or the
// connection fails (-1).
int dtfa_ps_serve(void* core, int fd, void* hdr_out) {
  PsCore* c = (PsCore*)core;
  std::vector<float> gbuf;
  for (;;) {
    Header h;
    if (!recv_all(fd, &h, sizeof(h))) return 0;
    if (memcmp(h.magic, "DTPS", 4) != 0) return -1;
    if (h.op != OP_PULL && h.op != OP_PUSH) {
      memcpy(hdr_out, &h, sizeof(h));
      return 1;
    }
    const int worker = h.worker;
    const long long clock = h.clock;
    bool init;
    long long gs;
    {
      std::lock_guard<std::mutex> lk(c->mu);
      init = c->initialized;
      gs = c->global_step;
    }
    if (c->stop.load()) {   // shutting down: tell the worker instead of dropping the connection
      if (h.nbytes) {
        gbuf.resize((h.nbytes + 3) / 4);
        if (!recv_all(fd, gbuf.data(), h.nbytes)) return 0;
      }
      if (!send_frame(fd, h.op, worker, ST_STOPPED, gs, nullptr, 0)) return -1;
      continue;
    }
    if (h.op == OP_PULL) {
      if (h.nbytes) {
        gbuf.resize((h.nbytes + 3) / 4);
        if (!recv_all(fd, gbuf.data(), h.nbytes)) return 0;
      }
      if (!init) {
        if (!send_frame(fd, OP_PULL, worker, ST_NOT_INIT, 0, nullptr, 0)) return -1;
        continue;
      }
      wait_for_staleness(c, worker, clock);
      {
        std::lock_guard<std::mutex> lk(c->mu);
        c->last_seen[worker] = now_s();
        gs = c->global_step;
      }
      if (c->stop.load()) {
        if (!send_frame(fd, OP_PULL, worker, ST_STOPPED, gs, nullptr, 0)) return -1;
        continue;
      }
      if (!send_frame(fd, OP_PULL, worker, ST_OK, gs, c->w, c->n * 4)) return -1;
      continue;
    }
    // PUSH
    gbuf.resize((h.nbytes + 3) / 4);
    if (h.nbytes && !recv_all(fd, gbuf.data(), h.nbytes)) return 0;
    if (!init) {
      if (!send_frame(fd, OP_PUSH, worker, ST_NOT_INIT, 0, nullptr, 0)) return -1;
      continue;
    }
    if (h.nbytes != c->n * 4) {
      if (!send_frame(fd, OP_PUSH, worker, ST_ERROR, 0, nullptr, 0)) return -1;
      continue;
    }
    long long step = 0;
    bool rejected = false;
    {
      std::lock_guard<std::mutex> lk(c->mu);
      if (c->debug) {
        auto it = c->clocks.find(worker);
        if (it != c->clocks.end() && it->second != clock) {
          c->violations.push_back("worker " + std::to_string(worker) + " pushed clock " + std::to_string(clock) +
                                  ", ps holds " + std::to_string(it->second));
          rejected = true;
        } else {
          c->clock_advances += 1;
          if (c->clock_advances != c->apply_count + 1)
            c->violations.push_back("applies " + std::to_string(c->apply_count + 1) + " != clock advances " +
                                    std::to_string(c->clock_advances));
        }
      }
      if (!rejected) {
        // the step this gradient is applied as (position in the schedule, Adam's t): one number per arrival
        step = c->global_step;
        c->global_step += 1;
        c->apply_count += 1;
      }
    }
    if (rejected) {
      if (!send_frame(fd, OP_PUSH, worker, ST_ERROR, 0, nullptr, 0)) return -1;
      continue;
    }
    apply_gradient(c, gbuf.data(), step);   // (outside the control mutex: shard mutexes only)
    {
      std::lock_guard<std::mutex> lk(c->mu);
      c->clocks[worker] = clock + 1;
      c->last_seen[worker] = now_s();
      gs = c->global_step;
      c->cv.notify_all();
    }
    if (!send_frame(fd, OP_PUSH, worker, ST_OK, gs, nullptr, 0)) return -1;
  }
}

}  // extern "C"
