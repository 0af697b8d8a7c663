// TEST INFRASTRUCTURE -- the kernels' certified screen (ilqr_amd/csrc/eig_screen.hpp) compiled for the host, so that
// its verdicts can be scored against LAPACK's on millions of matrices (tests/test_eig_screen.py).
//   g++ -O2 -ffp-contract=off -shared -fPIC -o libscreen_check.so screen_check.cpp
#include <stdint.h>
#include <string.h>

#include "../../ilqr_amd/csrc/eig_screen.hpp"

namespace {
struct RowMajor {
    const double* p;
    int w;
    double operator()(int i, int j) const { return p[i * w + j]; }
};
template <int W>
int run(const double* Q) { return ilqr::eig::screen<W>(RowMajor{Q, W}); }

int dispatch(const double* Q, int w)
{
    switch (w) {
    case 2: return run<2>(Q);
    case 3: return run<3>(Q);
    case 4: return run<4>(Q);
    case 5: return run<5>(Q);
    case 6: return run<6>(Q);
    case 7: return run<7>(Q);
    case 8: return run<8>(Q);
    case 16: return run<16>(Q);
    }
    return 2;
}

// tallies of the observer: [class of max|Q|: <1e5, <1e8, <1e12, <1e16, beyond][agrees, WRONG, in doubt]
long g_tally[5][3];
double g_closest[2];     // (not used by the verdict) reserved
}  // namespace

extern "C" {

// verdict[b] = screen of the row-major w x w matrix b: 0 none, 1 negative, 2 in doubt
void screen_verdicts(int w, const double* Q, long B, int32_t* verdict)
{
#pragma omp parallel for schedule(static)
    for (long b = 0; b < B; ++b) verdict[b] = dispatch(Q + b * w * w, w);
}

void screen_tally_reset(void) { memset(g_tally, 0, sizeof(g_tally)); }
void screen_tally_get(long* out15) { memcpy(out15, g_tally, sizeof(g_tally)); }

// observer for ilqr_oracle_set_eig_observer: scores the screen against the verdict the checker's LAPACK path gave
void screen_observer(const double* Q, int w, int verdict)
{
    double mq = 0.0;
    for (int e = 0; e < w * w; ++e) {
        const double a = fabs(Q[e]);
        if (a > mq) mq = a;
    }
    const int cls = mq < 1e5 ? 0 : mq < 1e8 ? 1 : mq < 1e12 ? 2 : mq < 1e16 ? 3 : 4;
    const int s = dispatch(Q, w);
    const int what = (s == 2) ? 2 : (s == verdict ? 0 : 1);
#pragma omp atomic
    g_tally[cls][what] += 1;
}

}  // extern "C"
