/* TEST INFRASTRUCTURE ONLY: stand-in for the generated <jxl/jxl_cms_export.h>. */
#ifndef JXL_CMS_EXPORT_H
#define JXL_CMS_EXPORT_H
#define JXL_CMS_EXPORT __attribute__((visibility("default")))
#define JXL_CMS_NO_EXPORT
#endif
