/*
 * jxl_threads_hip.h -- the JxlParallelRunner of the MI355X back-end.
 *
 * libjxl_threads_hip.so exports the SAME nine symbols as libjxl_threads (the four of the
 * thread-pool runner, the five of the resizable runner)
 *
 * (lib/include/jxl/thread_parallel_runner.h:45-66, implemented in
 * lib/threads/thread_parallel_runner.cc:68-109), with the same contract
 * (lib/include/jxl/parallel_runner.h:105-129): init(opaque, num_threads) is
 * called once on the calling thread and its non-zero result is returned;
 * func(opaque, i, thread_id) is called exactly once for every i in
 * [start_range, end_range) with thread_id < num_threads; the call blocks; the
 * runner is not re-entrant.  An unmodified djxl / JxlDecoder user links it in
 * place of libjxl_threads.
 *
 * The workers are plain host threads running the group tasks (entropy decode ->
 * jxlhip_submit_group); the HIP streams and the pinned staging those tasks'
 * uploads use belong to the jxlhip context, which therefore works under any
 * JxlParallelRunner.  (Round 1 gave every worker a stream of its own and exported
 * it through an extension symbol nothing consumed; both are gone.)  The
 * declarations below mirror the reference's so this header can be used without
 * libjxl's headers.
 */
#ifndef JXL_THREADS_HIP_H_
#define JXL_THREADS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define JXL_THREADS_HIP_EXPORT __attribute__((visibility("default")))
#else
#define JXL_THREADS_HIP_EXPORT
#endif

#ifndef JXL_PARALLEL_RUNNER_H_
typedef int JxlParallelRetCode;
#define JXL_PARALLEL_RET_SUCCESS (0)
#define JXL_PARALLEL_RET_RUNNER_ERROR (-1)
typedef JxlParallelRetCode (*JxlParallelRunInit)(void* jpegxl_opaque, size_t num_threads);
typedef void (*JxlParallelRunFunction)(void* jpegxl_opaque, uint32_t value, size_t thread_id);
#endif

#ifndef JXL_MEMORY_MANAGER_H_
/* lib/include/jxl/memory_manager.h:45-63 */
typedef void* (*jpegxl_alloc_func)(void* opaque, size_t size);
typedef void (*jpegxl_free_func)(void* opaque, void* address);
typedef struct JxlMemoryManagerStruct {
  void* opaque;
  jpegxl_alloc_func alloc;
  jpegxl_free_func free;
} JxlMemoryManager;
#endif

/* lib/threads/thread_parallel_runner.cc:68-75 */
JXL_THREADS_HIP_EXPORT JxlParallelRetCode JxlThreadParallelRunner(
    void* runner_opaque, void* jpegxl_opaque, JxlParallelRunInit init,
    JxlParallelRunFunction func, uint32_t start_range, uint32_t end_range);
/* :78-97 -- memory_manager: both callbacks NULL => malloc/free, exactly one
 * NULL => NULL is returned (lib/threads/thread_parallel_runner.cc:37-53) */
JXL_THREADS_HIP_EXPORT void* JxlThreadParallelRunnerCreate(
    const JxlMemoryManager* memory_manager, size_t num_worker_threads);
/* :99-105 */
JXL_THREADS_HIP_EXPORT void JxlThreadParallelRunnerDestroy(void* runner_opaque);
/* :107-109 */
JXL_THREADS_HIP_EXPORT size_t JxlThreadParallelRunnerDefaultNumWorkerThreads(void);

/* lib/include/jxl/resizable_parallel_runner.h:46-69 (lib/threads/resizable_parallel_runner.cc:172-195): the
 * runner whose thread count is set after the image size is known.  Same contract as the reference's: SetThreads(n)
 * keeps n - 1 workers and the calling thread is thread 0 of every run; a single task runs inline with
 * init(opaque, 1); SuggestThreads = min(hardware threads, xsize * ysize / 65536). */
JXL_THREADS_HIP_EXPORT JxlParallelRetCode JxlResizableParallelRunner(
    void* runner_opaque, void* jpegxl_opaque, JxlParallelRunInit init,
    JxlParallelRunFunction func, uint32_t start_range, uint32_t end_range);
JXL_THREADS_HIP_EXPORT void* JxlResizableParallelRunnerCreate(const JxlMemoryManager* memory_manager);
JXL_THREADS_HIP_EXPORT void JxlResizableParallelRunnerSetThreads(void* runner_opaque, size_t num_threads);
JXL_THREADS_HIP_EXPORT uint32_t JxlResizableParallelRunnerSuggestThreads(uint64_t xsize, uint64_t ysize);
JXL_THREADS_HIP_EXPORT void JxlResizableParallelRunnerDestroy(void* runner_opaque);

#ifdef __cplusplus
}
#endif
#endif
