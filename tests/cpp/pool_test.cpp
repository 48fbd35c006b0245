// Host-only test of csrc/qn_pool.h (no HIP): fn(0) on the caller, fn(1..n-1) on parked workers, nested fan-out (a per-GPU worker fanning out over its contexts),
// concurrent callers, reuse of the parked threads across calls.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <thread>
#include <vector>
#include "qn_pool.h"

int main() {
  qn::WorkerPool& p = qn::WorkerPool::instance();
  // 1. plain fan-out, many calls: every index exactly once, index 0 on the calling thread
  for (int rep = 0; rep < 200; rep++) {
    const uint32_t n = 1 + rep % 7;
    std::vector<std::atomic<int>> hits(n);
    for (auto& h : hits) h = 0;
    std::thread::id caller = std::this_thread::get_id(); bool zero_on_caller = false;
    p.run(n, [&](uint32_t i) { hits[i]++; if (i == 0) zero_on_caller = std::this_thread::get_id() == caller; });
    for (uint32_t i = 0; i < n; i++) if (hits[i] != 1) { printf("FAIL: index %u ran %d times\n", i, (int)hits[i]); return 1; }
    if (!zero_on_caller) { printf("FAIL: fn(0) did not run on the caller\n"); return 1; }
  }
  const uint32_t after_plain = p.threads();
  if (after_plain > 6) { printf("FAIL: %u threads for fan-outs of at most 7\n", after_plain); return 1; }
  // 2. nested: 4 "GPU workers", each fanning out over 3 "contexts"; tasks block on each other's completion inside (every queued task needs a thread of its own)
  std::atomic<int> total{0};
  p.run(4, [&](uint32_t g) {
    std::atomic<int> inner{0};
    p.run(3, [&](uint32_t c) { inner += (int)(10 * g + c); std::this_thread::sleep_for(std::chrono::milliseconds(2)); });
    total += inner;
  });
  if (total != (0 + 1 + 2) * 4 + 10 * 3 * (0 + 1 + 2 + 3)) { printf("FAIL: nested sum %d\n", (int)total); return 1; }
  // 3. concurrent callers from plain std::threads
  std::atomic<int> c2{0};
  std::vector<std::thread> th;
  for (int t = 0; t < 4; t++) th.emplace_back([&] { for (int r = 0; r < 50; r++) p.run(3, [&](uint32_t) { c2++; }); });
  for (auto& t : th) t.join();
  if (c2 != 4 * 50 * 3) { printf("FAIL: concurrent callers %d\n", (int)c2); return 1; }
  // 4. n = 0 and n = 1 never touch the pool
  int ran = 0; p.run(0, [&](uint32_t) { ran++; }); p.run(1, [&](uint32_t i) { ran += 1 + (int)i; });
  if (ran != 1) { printf("FAIL: degenerate fan-outs\n"); return 1; }
  // 5. a task that throws (pool thread or caller): every other task still runs, run() returns only after all of them, the exception reaches the caller, the pool lives on
  for (uint32_t bad : {0u, 2u}) {
    std::atomic<int> done{0}; bool caught = false;
    try { p.run(4, [&](uint32_t i) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); if (i == bad) throw std::runtime_error("boom"); done++; }); }
    catch (const std::runtime_error&) { caught = true; }
    if (!caught || done != 3) { printf("FAIL: exception path (bad %u): caught %d done %d\n", bad, (int)caught, (int)done); return 1; }
  }
  std::atomic<int> c5{0}; p.run(5, [&](uint32_t) { c5++; });
  if (c5 != 5) { printf("FAIL: pool unusable after an exception\n"); return 1; }
  printf("ok: %u parked threads\n", p.threads());
  return 0;
}
