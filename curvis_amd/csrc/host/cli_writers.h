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
    cv_space_.wait(g, [this] { return q_.size() < 64; }); /* bound the frames held in host memory */
    q_.push_back(std::move(job));
    cv_work_.notify_one();
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
  PinnedPool(size_t bytes_each, int n) : bytes_(bytes_each) {
    for (int i = 0; i < n; ++i) {
      void *p = nullptr;
      if (curvis_host_alloc(bytes_each, &p) != CURVIS_OK || !p) break;
      free_.push_back((uint8_t *)p);
      all_.push_back((uint8_t *)p);
    }
  }
  ~PinnedPool() {
    for (uint8_t *p : all_) curvis_host_free(p);
  }
  size_t buffers() const { return all_.size(); }
  /* a buffer that returns to the pool when the last holder lets go of it */
  std::shared_ptr<uint8_t> take(double *waited_s) {
    std::unique_lock<std::mutex> g(mu_);
    const double t0 = pngio::now_s();
    cv_.wait(g, [this] { return !free_.empty(); });
    if (waited_s) *waited_s += pngio::now_s() - t0;
    uint8_t *p = free_.back();
    free_.pop_back();
    return std::shared_ptr<uint8_t>(p, [this](uint8_t *q) {
      {
        std::lock_guard<std::mutex> g2(mu_);
        free_.push_back(q);
      }
      cv_.notify_one();
    });
  }

 private:
  size_t bytes_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<uint8_t *> free_, all_;
};

}  // namespace

#endif /* CURVIS_CLI_WRITERS_H */
