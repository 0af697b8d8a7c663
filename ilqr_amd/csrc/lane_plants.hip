// run_part for the car and for user-defined plants; see pipeline_host.hpp for why
// this is a translation unit of its own.
#include "pipeline_host.hpp"

namespace ilqr {
#ifndef ILQR_ONLY_USER
template int run_part<CarModel, false>(Pipe, hipStream_t, ilqr_timing*);
#endif
#if defined(ILQR_USER_MODEL) && !defined(USER_WIDE)
template int run_part<UserModel, false>(Pipe, hipStream_t, ilqr_timing*);
#endif
}  // namespace ilqr
