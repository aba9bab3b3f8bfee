// pnx_internal.hpp -- shared by the translation units of libpnx_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/pnx.h"

namespace pnx {
struct CurvefitArgs;
// records a printf-style message for pnx_last_error() and returns `code`
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// Configuration through the environment, two classes (include/pnx.h, "Environment"):
//   env_int      documented settings of a production process (cache sizes, helper thread counts);
//   dev_getenv / dev_env_int   developer switches -- A/B kernel selection, chunk sizes of the host pipelines, trace output, the
//                rejection test hook of the NNLS block kernel: read only when the process was started with PNX_ENABLE_TEST_HOOKS=1
//                (looked at once, at the first query); otherwise they answer "not set", whatever the environment holds.
int env_int(const char *name, int dflt, int lo, int hi);
const char *dev_getenv(const char *name);
int dev_env_int(const char *name, int dflt, int lo, int hi);
}  // namespace pnx

// the seven curve-fit translation units (one model each): internal linkage names, not exported (-fvisibility=hidden)
extern "C" {
int pnx_launch_curvefit_m0(int, int, const pnx::CurvefitArgs *, int, void *);
int pnx_launch_curvefit_m1(int, int, const pnx::CurvefitArgs *, int, void *);
int pnx_launch_curvefit_m2(int, int, const pnx::CurvefitArgs *, int, void *);
int pnx_launch_curvefit_m3(int, int, const pnx::CurvefitArgs *, int, void *);
int pnx_launch_curvefit_m4(int, int, const pnx::CurvefitArgs *, int, void *);
int pnx_launch_curvefit_m5(int, int, const pnx::CurvefitArgs *, int, void *);
int pnx_launch_curvefit_m6(int, int, const pnx::CurvefitArgs *, int, void *);
}
