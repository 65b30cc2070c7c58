/* TEST INFRASTRUCTURE ONLY: stand-in for the cmake-generated <jxl/jxl_export.h>
 * (oracle/_ref build of the libjxl reference, see oracle/build_ref.py). */
#ifndef JXL_EXPORT_H
#define JXL_EXPORT_H
#define JXL_EXPORT __attribute__((visibility("default")))
#define JXL_NO_EXPORT
#define JXL_DEPRECATED __attribute__((deprecated))
#endif
