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

#ifdef ILQR_DEBUG_COUNTERS
// diagnostics build only: the counters of solver_device.hpp as this translation unit's kernels left them
extern "C" int ilqr_debug_counters(unsigned long long* out16, int reset)
{
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(ilqr::g_dbg), sizeof(ilqr::g_dbg)) != hipSuccess) return -3;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ilqr::g_dbg), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
extern "C" int ilqr_debug_eig_cycles(unsigned long long* out8, int reset)
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(ilqr::eig::g_eig_cycles), sizeof(ilqr::eig::g_eig_cycles)) != hipSuccess) return -3;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ilqr::eig::g_eig_cycles), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
extern "C" int ilqr_debug_eig_trips(unsigned long long* out12, int reset)
{
    if (out12 && hipMemcpyFromSymbol(out12, HIP_SYMBOL(ilqr::eig::g_eig_trips), sizeof(ilqr::eig::g_eig_trips)) != hipSuccess) return -3;
    if (reset) {
        unsigned long long z[12] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ilqr::eig::g_eig_trips), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
extern "C" int ilqr_debug_histograms(unsigned long long* out32, int reset)
{
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(ilqr::g_hist), sizeof(ilqr::g_hist)) != hipSuccess) return -3;
    if (reset) {
        unsigned long long z[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ilqr::g_hist), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
#endif
