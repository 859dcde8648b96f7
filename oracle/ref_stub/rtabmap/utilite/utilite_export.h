/* Stand-in for the CMake-generated utilite_export.h (the reference generates it at configure time;
 * rtflann/util/heap.h:36 pulls ULogger.h which needs these macros).  Test infrastructure only. */
#ifndef UTILITE_EXPORT_H
#define UTILITE_EXPORT_H
#define UTILITE_EXPORT
#define UTILITE_NO_EXPORT
#define UTILITE_DEPRECATED
#endif
