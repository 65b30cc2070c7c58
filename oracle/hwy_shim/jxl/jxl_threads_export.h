/* TEST INFRASTRUCTURE ONLY: stand-in for the generated <jxl/jxl_threads_export.h>. */
#ifndef JXL_THREADS_EXPORT_H
#define JXL_THREADS_EXPORT_H
#define JXL_THREADS_EXPORT __attribute__((visibility("default")))
#define JXL_THREADS_NO_EXPORT
#endif
