// pnx_internal.hpp -- shared by the translation units of libpnx_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/pnx.h"

namespace pnx {
struct CurvefitArgs;
// records a printf-style message for pnx_last_error() and returns `code`
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace pnx

extern "C" {
int pnx_launch_curvefit_m0(int, int, const pnx::CurvefitArgs *, int, void *);
int pnx_launch_curvefit_m1(int, int, const pnx::CurvefitArgs *, int, void *);
int pnx_launch_curvefit_m2(int, int, const pnx::CurvefitArgs *, int, void *);
int pnx_launch_curvefit_m3(int, int, const pnx::CurvefitArgs *, int, void *);
int pnx_launch_curvefit_m4(int, int, const pnx::CurvefitArgs *, int, void *);
int pnx_launch_curvefit_m5(int, int, const pnx::CurvefitArgs *, int, void *);
int pnx_launch_curvefit_m6(int, int, const pnx::CurvefitArgs *, int, void *);
}
