// Project-owned fixed-size thread pool with the two members the reference uses
// (ctc_beam_search_decoder.cpp:259,266): ThreadPool(size_t) and
// enqueue(f, args...) -> std::future<result>.  The reference's ThreadPool
// submodule (progschj/ThreadPool, unpinned) is an empty directory.  Scheduling
// only: no arithmetic happens here.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

class ThreadPool {
 public:
  explicit ThreadPool(std::size_t n) {
    for (std::size_t i = 0; i < n; ++i) workers_.emplace_back([this] { run(); });
  }
  ~ThreadPool() {
    {
      std::lock_guard<std::mutex> g(mu_);
      closing_ = true;
    }
    cv_.notify_all();
    for (auto &w : workers_) w.join();
  }
  template <class F, class... A>
  auto enqueue(F &&f, A &&...a) -> std::future<typename std::result_of<F(A...)>::type> {
    using R = typename std::result_of<F(A...)>::type;
    auto job = std::make_shared<std::packaged_task<R()>>(std::bind(std::forward<F>(f), std::forward<A>(a)...));
    std::future<R> fut = job->get_future();
    {
      std::lock_guard<std::mutex> g(mu_);
      queue_.emplace_back([job] { (*job)(); });
    }
    cv_.notify_one();
    return fut;
  }

 private:
  void run() {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return closing_ || !queue_.empty(); });
        if (queue_.empty()) return;
        job = std::move(queue_.front());
        queue_.pop_front();
      }
      job();
    }
  }
  std::vector<std::thread> workers_;
  std::deque<std::function<void()>> queue_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool closing_ = false;
};
