// Micro-benchmark: issue rate of fp64 vector instructions on one SIMD as a function of
// resident waves and of the number of independent dependency chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH, int OP>
__global__ void chains(double* out, int iters, double a, double b)
{
    double v[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) v[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (OP == 0) v[c] = __builtin_fma(v[c], a, b);
            else if (OP == 1) v[c] = v[c] * a;
            else v[c] = v[c] + b;
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += v[c];
    if (s == 12345.0) out[0] = s;
}

template <int CH, int OP>
void run(const char* name, int waves_per_simd)
{
    double* out; hipMalloc(&out, 8);
    const int iters = 20000;
    // one workgroup of 64 threads per wave; grid sized to put `waves_per_simd` waves on every SIMD (256 CUs x 4)
    const int blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((chains<CH, OP>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    double instr_per_wave = (double)iters * CH;
    double ns_per_instr_wave = ms * 1e6 / instr_per_wave;                  // latency seen by one wave
    double ns_per_instr_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
    printf("%-6s chains %2d waves/SIMD %d: %8.3f ms  %6.2f ns per instr per wave, %6.2f ns per instr per SIMD  (%.1f TFLOP/s if fma)\n",
           name, CH, waves_per_simd, ms, ns_per_instr_wave, ns_per_instr_simd,
           2.0 * 64 * instr_per_wave * blocks / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4, 8}) { run<1, 0>("fma", w); run<2, 0>("fma", w); run<4, 0>("fma", w); run<8, 0>("fma", w); }
    for (int w : {1, 2}) { run<1, 1>("mul", w); run<4, 1>("mul", w); run<1, 2>("add", w); run<4, 2>("add", w); }
    return 0;
}
