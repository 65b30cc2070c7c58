// runner.cc -- libjxl_threads_hip.so: the JxlParallelRunner shipped with the back-end (see
// include/jxl_threads_hip.h).  Replaces lib/threads/thread_parallel_runner{.cc,_internal.cc} and
// resizable_parallel_runner.cc behind the same nine C symbols.  Plain host threads: the streams and the pinned
// staging that the group tasks' uploads use belong to the jxlhip context (context.hip: jxlhip_submit_group picks
// a stream of its pool per call), so that the back-end works under ANY JxlParallelRunner, this one included.
//
// Scheduling: one shared atomic cursor over [begin, end); every participant
// claims a chunk of max(1, remaining / (4 * threads)) tasks per grab, so early
// chunks are large and the tail is fine-grained.  With zero workers the calling
// thread runs everything as thread 0.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/jxl_threads_hip.h"

namespace {

struct Runner {
  JxlMemoryManager mm{};
  size_t num_workers = 0;
  std::vector<std::thread> threads;

  std::mutex mu;
  std::condition_variable cv_start, cv_done;
  uint64_t epoch = 0;  // bumped for every Run
  bool quit = false;
  size_t running = 0;  // workers still inside the current epoch

  // current job
  void* opaque = nullptr;
  JxlParallelRunFunction func = nullptr;
  uint32_t end = 0;
  std::atomic<uint32_t> next{0};
  std::atomic<bool> in_run{false};

  void Drain(size_t thread_id) {
    const size_t nthreads = num_workers ? num_workers : 1;
    for (;;) {
      uint32_t cur = next.load(std::memory_order_relaxed);
      uint32_t take;
      do {
        if (cur >= end) return;
        const uint32_t remaining = end - cur;
        take = remaining / (uint32_t)(4 * nthreads);
        if (take < 1) take = 1;
      } while (!next.compare_exchange_weak(cur, cur + take, std::memory_order_relaxed));
      for (uint32_t i = cur; i < cur + take; i++) func(opaque, i, thread_id);
    }
  }

  void WorkerMain(size_t thread_id) {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lock(mu);
        cv_start.wait(lock, [&] { return quit || epoch != seen; });
        if (quit) return;
        seen = epoch;
      }
      Drain(thread_id);
      {
        std::lock_guard<std::mutex> lock(mu);
        if (--running == 0) cv_done.notify_all();
      }
    }
  }
};

void* MMAlloc(const JxlMemoryManager& mm, size_t n) {
  return mm.alloc ? mm.alloc(mm.opaque, n) : malloc(n);
}
void MMFree(const JxlMemoryManager& mm, void* p) {
  if (mm.free) mm.free(mm.opaque, p);
  else free(p);
}

}  // namespace

extern "C" {

JxlParallelRetCode JxlThreadParallelRunner(void* runner_opaque, void* jpegxl_opaque,
                                           JxlParallelRunInit init, JxlParallelRunFunction func,
                                           uint32_t start_range, uint32_t end_range) {
  Runner* r = static_cast<Runner*>(runner_opaque);
  if (!r || start_range > end_range) return JXL_PARALLEL_RET_RUNNER_ERROR;
  if (start_range == end_range) return JXL_PARALLEL_RET_SUCCESS;
  bool expected = false;
  if (!r->in_run.compare_exchange_strong(expected, true))
    return JXL_PARALLEL_RET_RUNNER_ERROR;  // not re-entrant
  const size_t nthreads = r->num_workers ? r->num_workers : 1;
  const JxlParallelRetCode rc = init(jpegxl_opaque, nthreads);
  if (rc != 0) {
    r->in_run.store(false);
    return rc;
  }
  r->opaque = jpegxl_opaque;
  r->func = func;
  r->end = end_range;
  r->next.store(start_range, std::memory_order_relaxed);
  if (r->num_workers == 0) {
    r->Drain(0);
  } else {
    {
      std::lock_guard<std::mutex> lock(r->mu);
      r->running = r->num_workers;
      r->epoch++;
    }
    r->cv_start.notify_all();
    std::unique_lock<std::mutex> lock(r->mu);
    r->cv_done.wait(lock, [&] { return r->running == 0; });
  }
  r->in_run.store(false);
  return JXL_PARALLEL_RET_SUCCESS;
}

void* JxlThreadParallelRunnerCreate(const JxlMemoryManager* memory_manager,
                                    size_t num_worker_threads) {
  JxlMemoryManager mm{};
  if (memory_manager) {
    mm = *memory_manager;
    if ((mm.alloc == nullptr) != (mm.free == nullptr)) return nullptr;
  }
  void* mem = MMAlloc(mm, sizeof(Runner));
  if (!mem) return nullptr;
  Runner* r = new (mem) Runner();
  r->mm = mm;
  r->num_workers = num_worker_threads;
  r->threads.reserve(num_worker_threads);
  for (size_t i = 0; i < num_worker_threads; i++)
    r->threads.emplace_back([r, i] { r->WorkerMain(i); });
  return r;
}

void JxlThreadParallelRunnerDestroy(void* runner_opaque) {
  Runner* r = static_cast<Runner*>(runner_opaque);
  if (!r) return;
  {
    std::lock_guard<std::mutex> lock(r->mu);
    r->quit = true;
  }
  r->cv_start.notify_all();
  for (auto& t : r->threads) t.join();
  const JxlMemoryManager mm = r->mm;
  r->~Runner();
  MMFree(mm, r);
}

size_t JxlThreadParallelRunnerDefaultNumWorkerThreads(void) {
  return std::thread::hardware_concurrency();
}

}  // extern "C"

// ---- JxlResizableParallelRunner (lib/include/jxl/resizable_parallel_runner.h:46-69,
// lib/threads/resizable_parallel_runner.cc:26-195): the runner djxl-like tools use when the number of
// threads is only known once the image size is (JxlResizableParallelRunnerSuggestThreads).  Contract
// of the reference: SetThreads(n) keeps n - 1 workers, the CALLING thread is thread 0 of every run;
// init gets min(workers + 1, tasks) (1 for a single task, which runs inline).
namespace {

struct ResizableRunner {
  JxlMemoryManager mm{};
  std::mutex mu;
  std::condition_variable cv_start, cv_done;
  std::vector<std::thread> workers;
  size_t desired = 0;     // workers that should exist
  uint64_t epoch = 0;
  size_t participants = 0;  // workers [0, participants) take part in the current run
  size_t running = 0;
  void* opaque = nullptr;
  JxlParallelRunFunction func = nullptr;
  uint32_t end = 0;
  std::atomic<uint32_t> next{0};

  void Drain(size_t thread_id) {
    for (;;) {
      const uint32_t t = next.fetch_add(1, std::memory_order_relaxed);
      if (t >= end) return;
      func(opaque, t, thread_id);
    }
  }
  // seen: the epoch at the moment SetThreads CREATED this worker (read by the creating thread): runs that
  // happened before it existed are not its to join (it was never counted in `running`), every run started
  // afterwards counts it -- also when its thread only gets going after that run began
  void WorkerMain(size_t id, uint64_t seen) {
    for (;;) {
      {
        std::unique_lock<std::mutex> lock(mu);
        cv_start.wait(lock, [&] { return id >= desired || (epoch != seen && id < participants); });
        if (id >= desired) return;
        seen = epoch;
      }
      Drain(id + 1);
      {
        std::lock_guard<std::mutex> lock(mu);
        if (--running == 0) cv_done.notify_all();
      }
    }
  }
  void SetThreads(size_t num) {
    if (num > 0) num -= 1;
    uint64_t now;
    {
      std::lock_guard<std::mutex> lock(mu);
      desired = num;
      now = epoch;
    }
    cv_start.notify_all();
    for (size_t i = workers.size(); i < num; i++) workers.emplace_back([this, i, now] { WorkerMain(i, now); });
    if (workers.size() > num) {
      for (size_t i = num; i < workers.size(); i++) workers[i].join();
      workers.resize(num);
    }
  }
};

}  // namespace

extern "C" {

JxlParallelRetCode JxlResizableParallelRunner(void* runner_opaque, void* jpegxl_opaque, JxlParallelRunInit init,
                                              JxlParallelRunFunction func, uint32_t start_range,
                                              uint32_t end_range) {
  ResizableRunner* r = static_cast<ResizableRunner*>(runner_opaque);
  if (!r || start_range > end_range) return JXL_PARALLEL_RET_RUNNER_ERROR;
  if (start_range == end_range) return JXL_PARALLEL_RET_SUCCESS;
  if (start_range + 1 == end_range) {
    const JxlParallelRetCode rc = init(jpegxl_opaque, 1);
    if (rc != 0) return rc;
    func(jpegxl_opaque, start_range, 0);
    return rc;
  }
  const size_t tasks = end_range - start_range;
  const size_t n = r->workers.size() + 1 < tasks ? r->workers.size() + 1 : tasks;
  const JxlParallelRetCode rc = init(jpegxl_opaque, n);
  if (rc != 0) return rc;
  {
    std::lock_guard<std::mutex> lock(r->mu);
    r->opaque = jpegxl_opaque;
    r->func = func;
    r->end = end_range;
    r->next.store(start_range, std::memory_order_relaxed);
    r->participants = n - 1;
    r->running = n - 1;
    r->epoch++;
  }
  r->cv_start.notify_all();
  r->Drain(0);
  std::unique_lock<std::mutex> lock(r->mu);
  r->cv_done.wait(lock, [&] { return r->running == 0; });
  r->participants = 0;  // nobody may join this run any more (a worker that had not woken up yet stays asleep)
  return JXL_PARALLEL_RET_SUCCESS;
}

void* JxlResizableParallelRunnerCreate(const JxlMemoryManager* memory_manager) {
  JxlMemoryManager mm{};
  if (memory_manager) {
    mm = *memory_manager;
    if ((mm.alloc == nullptr) != (mm.free == nullptr)) return nullptr;
  }
  void* mem = MMAlloc(mm, sizeof(ResizableRunner));
  if (!mem) return nullptr;
  ResizableRunner* r = new (mem) ResizableRunner();
  r->mm = mm;
  return r;
}

void JxlResizableParallelRunnerSetThreads(void* runner_opaque, size_t num_threads) {
  if (runner_opaque) static_cast<ResizableRunner*>(runner_opaque)->SetThreads(num_threads);
}

uint32_t JxlResizableParallelRunnerSuggestThreads(uint64_t xsize, uint64_t ysize) {
  // ~one thread per group (resizable_parallel_runner.cc:189-194)
  const uint64_t groups = xsize * ysize / (256 * 256);
  const uint64_t hw = std::thread::hardware_concurrency();
  return (uint32_t)(groups < hw ? groups : hw);
}

void JxlResizableParallelRunnerDestroy(void* runner_opaque) {
  ResizableRunner* r = static_cast<ResizableRunner*>(runner_opaque);
  if (!r) return;
  r->SetThreads(0);
  const JxlMemoryManager mm = r->mm;
  r->~ResizableRunner();
  MMFree(mm, r);
}

}  // extern "C"
