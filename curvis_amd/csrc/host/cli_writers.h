/* cli_writers.h -- part of the `curvis` binary (host/curvis_cli.cpp includes the parts in order; one translation unit):
 * host side of the frame output: PNG writer pool and page-locked batch buffers. */
#ifndef CURVIS_CLI_WRITERS_H
#define CURVIS_CLI_WRITERS_H

namespace {

/* std::fs::remove_dir_all (src/rendering.rs:278): no shell involved, the path is never interpreted */
int rm_rf(const std::string &dir) {
  std::error_code ec;
  std::filesystem::remove_all(std::filesystem::path(dir), ec);
  return ec ? 1 : 0;
}

/* frame writers: PNG encoding (zlib) costs more host time per frame than the GPU needs to render it, so
 * frames are compressed and written by a small pool of host threads while the GPU renders the next batch. */
class WriterPool {
 public:
  explicit WriterPool(int n) {
    for (int i = 0; i < n; ++i) th_.emplace_back([this] { run(); });
  }
  ~WriterPool() { finish(); }
  void submit(std::function<void()> job) {
    std::unique_lock<std::mutex> g(mu_);
    cv_space_.wait(g, [this] { return q_.size() < bound_; }); /* bound the frames held in host memory */
    q_.push_back(std::move(job));
    cv_work_.notify_one();
  }
  /* Jobs that share page-locked BATCH buffers hold no memory of their own -- the buffer pool bounds that --, and a worker hands over
   * a whole batch at once: with the queue shorter than a batch (128 frames in efficient mode) it waits here for the writers to
   * make room.  Raised by such callers (cli_video.h says when); never lowered. */
  void raise_bound(size_t n) {
    std::lock_guard<std::mutex> g(mu_);
    if (n > bound_) bound_ = n;
    cv_space_.notify_all();
  }
  void finish() {
    {
      std::lock_guard<std::mutex> g(mu_);
      if (done_) return;
      done_ = true;
    }
    cv_work_.notify_all();
    for (auto &t : th_) t.join();
  }

 private:
  void run() {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_work_.wait(g, [this] { return done_ || !q_.empty(); });
        if (q_.empty()) return;
        job = std::move(q_.front());
        q_.pop_front();
        cv_space_.notify_one();
      }
      job();
    }
  }
  std::mutex mu_;
  size_t bound_ = 64;
  std::condition_variable cv_work_, cv_space_;
  std::deque<std::function<void()>> q_;
  std::vector<std::thread> th_;
  bool done_ = false;
};

/* Page-locked batch buffers of the video workers: the render call copies a batch of frames into one of them by DMA, the
 * writer threads encode straight out of it, and the last frame written gives it back -- no pageable bounce copy (2 GB/s)
 * and no per-frame memcpy between the render call and the encoder.  A worker that finds the pool empty waits: that is
 * the back-pressure of a host that cannot keep up. */
class PinnedPool {
 public:
  /* up to `n` page-locked buffers of `bytes_each`.  ONE is made here (so that the caller learns at once whether page-locked
   * memory can be had), the others when a taker finds none free: pinning costs ~0.2 ms per MB and releasing as much again, a
   * worker of a short run never needs its third buffer, and four contexts per GPU each held three of 200 MB for a 240-frame
   * video -- 0.4 s of a 1.3 s run went into pinning and un-pinning memory nobody wrote to.  resize() makes the buffers handed
   * out from then on larger (a batch of zlib streams that turned out not to fit); smaller ones are released as they return. */
  PinnedPool(size_t bytes_each, int n) : bytes_(bytes_each), max_(n) { (void)grow(); }
  ~PinnedPool() {
    for (auto &b : all_) curvis_host_free(b.first);
  }
  /* buffers this pool has or can still try to make (0: page-locked memory cannot be had at all) */
  size_t buffers() const {
    std::lock_guard<std::mutex> g(mu_); /* writer threads let go of buffers (and drop outgrown ones) at any time */
    return all_.empty() ? 0 : (size_t)max_;
  }
  size_t bytes_each() const { return bytes_; }
  void resize(size_t bytes_each) {
    std::lock_guard<std::mutex> g(mu_);
    if (bytes_each <= bytes_) return;
    bytes_ = bytes_each;
    for (uint8_t *p : free_) drop(p);
    free_.clear();
  }
  /* a buffer of bytes_each() that returns to the pool when the last holder lets go of it; empty when none can be had */
  std::shared_ptr<uint8_t> take(double *waited_s) {
    std::unique_lock<std::mutex> g(mu_);
    const double t0 = pngio::now_s();
    while (free_.empty()) {
      if ((int)all_.size() < max_) {
        if (grow()) break;
        if (all_.empty()) return std::shared_ptr<uint8_t>(); /* no page-locked memory (any more): the caller uses pageable memory */
        max_ = (int)all_.size();                             /* live with what there is */
      }
      cv_.wait(g);
    }
    if (waited_s) *waited_s += pngio::now_s() - t0;
    uint8_t *p = free_.back();
    free_.pop_back();
    return std::shared_ptr<uint8_t>(p, [this](uint8_t *q) {
      {
        std::lock_guard<std::mutex> g2(mu_);
        size_t sz = 0;
        for (auto &b : all_)
          if (b.first == q) sz = b.second;
        if (sz < bytes_)
          drop(q); /* made before a resize(): too small for what is handed out now */
        else
          free_.push_back(q);
      }
      cv_.notify_one();
    });
  }

 private:
  bool grow() {
    void *p = nullptr;
    if (curvis_host_alloc(bytes_, &p) != CURVIS_OK || !p) return false;
    free_.push_back((uint8_t *)p);
    all_.emplace_back((uint8_t *)p, bytes_);
    return true;
  }
  void drop(uint8_t *p) {
    for (size_t i = 0; i < all_.size(); ++i)
      if (all_[i].first == p) {
        all_.erase(all_.begin() + (long)i);
        break;
      }
    curvis_host_free(p);
  }
  size_t bytes_;
  int max_;
  mutable std::mutex mu_;
  std::condition_variable cv_;
  std::vector<uint8_t *> free_;
  std::vector<std::pair<uint8_t *, size_t>> all_;
};

}  // namespace

#endif /* CURVIS_CLI_WRITERS_H */
