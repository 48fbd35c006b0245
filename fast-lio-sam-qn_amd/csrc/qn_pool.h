// qn_pool.h - host worker threads that outlive a call (host code only, no kernels).
//
// qn_icp_alignment_batch / qn_coarse_to_fine_align_batch drive one context (= one hipStream) per worker and qn_multi_align_best one worker per GPU.
// A candidate batch of 8 pairs is ~2-3 ms of GPU work; creating and joining std::threads per call (what rounds 3-4 did) costs 40-90 us per thread on
// the calling thread - a few per cent of exactly the short calls the 8-GPU deployment makes (8 pairs per rank).  The pool keeps
// its threads parked on a condition variable between calls.  run(n, fn) executes fn(0) on the caller and fn(1..n-1) on pool threads and returns when
// all are done; calls may nest (a per-GPU worker calling the per-context fan-out): the pool grows so that every queued task has a thread of its own
// and never waits behind a task that is itself waiting - tasks here block on GPU streams, so "one thread per task in flight" is the right size.
// The pool is deliberately leaked (threads detached): a static destructor joining threads that sit in HIP calls at process exit is a known way to hang.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>

namespace qn {

class WorkerPool {
 public:
  static WorkerPool& instance() { static WorkerPool* p = new WorkerPool(); return *p; }

  // fn(i) for i in [0, n): i = 0 on the calling thread, the rest on pool threads; returns after every fn returned.  Whatever goes wrong - a thread cannot be created,
  // an fn throws (bad_alloc in a worker's std::vector, ...) - the queued tasks are all executed or withdrawn BEFORE run() returns (they point at this frame's latch and fn),
  // and the first exception is rethrown on the caller.  The pool never holds more than kMaxThreads threads: tasks beyond that run on the caller, in order.
  static constexpr uint32_t kMaxThreads = 256;
  void run(uint32_t n, const std::function<void(uint32_t)>& fn) {
    if (n <= 1) { if (n == 1) fn(0); return; }
    Latch latch; latch.left = n - 1;
    {
      std::unique_lock<std::mutex> lk(mu_);
      for (uint32_t i = 1; i < n; i++) queue_.push_back(Task{&fn, i, &latch});
      in_flight_ += n - 1;
      while (threads_ < in_flight_ && threads_ < kMaxThreads) {
        try { std::thread(&WorkerPool::loop, this).detach(); threads_++; }
        catch (...) { break; }                                          // no more threads to be had: the caller works the rest off below
      }
    }
    cv_.notify_all();
    std::exception_ptr err;
    try { fn(0); } catch (...) { err = std::current_exception(); }
    // tasks of THIS call that no pool thread has taken yet (too few threads, or every thread busy with a task that is itself waiting): run them here
    for (;;) {
      Task t; bool have = false;
      {
        std::unique_lock<std::mutex> lk(mu_);
        if (threads_ >= in_flight_) break;                              // every queued task has a thread coming for it
        for (auto it = queue_.begin(); it != queue_.end(); ++it) if (it->latch == &latch) { t = *it; queue_.erase(it); have = true; break; }
        if (have) in_flight_--;
      }
      if (!have) break;
      try { (*t.fn)(t.i); } catch (...) { if (!err) err = std::current_exception(); }
      { std::unique_lock<std::mutex> lk(latch.mu); --latch.left; }
    }
    {
      std::unique_lock<std::mutex> lk(latch.mu);
      latch.cv.wait(lk, [&] { return latch.left == 0; });
      if (!err && latch.err) err = latch.err;
    }
    if (err) std::rethrow_exception(err);
  }

  uint32_t threads() { std::unique_lock<std::mutex> lk(mu_); return threads_; }

 private:
  struct Latch { std::mutex mu; std::condition_variable cv; uint32_t left = 0; std::exception_ptr err; };
  struct Task { const std::function<void(uint32_t)>* fn; uint32_t i; Latch* latch; };

  void loop() {
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return !queue_.empty(); });
        t = queue_.front(); queue_.pop_front();
      }
      std::exception_ptr err;
      try { (*t.fn)(t.i); } catch (...) { err = std::current_exception(); }      // (an exception must not take the process down from a detached thread: it goes back to the caller of run())
      { std::unique_lock<std::mutex> lk(mu_); in_flight_--; }
      { std::unique_lock<std::mutex> lk(t.latch->mu); if (err && !t.latch->err) t.latch->err = err; if (--t.latch->left == 0) t.latch->cv.notify_all(); }      // (notify under the lock: the latch lives on the waiter's stack)
    }
  }

  std::mutex mu_; std::condition_variable cv_; std::deque<Task> queue_;
  uint32_t threads_ = 0, in_flight_ = 0;
};

}  // namespace qn
