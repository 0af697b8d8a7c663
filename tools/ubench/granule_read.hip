// Micro-benchmark: how fast can a wavefront stream per-lane contiguous granules?
//   mode 0: strided SoA, 8 B per lane per load (lane l reads base[i*stride + l])
//   mode G: granule of G doubles per lane (lane l reads G consecutive doubles at base[(i*n + l)*G]) with 16 B loads
// Dense (every lane active) and sparse (one lane in 8 active, random) variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

template <int G>
__global__ void rd(const double* __restrict__ p, long nslots, int rows, const int* __restrict__ slot_of, long nact, double* out)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nact) return;
    long s = slot_of ? slot_of[t] : t;
    double acc = 0.0;
    if (G == 0) {
        for (int i = 0; i < rows * 8; ++i) acc += p[(long)i * nslots + s];
    } else {
        for (int i = 0; i < rows * 8 / G; ++i) {
            const double2* q = (const double2*)(p + ((long)i * nslots + s) * G);
#pragma unroll
            for (int j = 0; j < G / 2; ++j) { double2 v = q[j]; acc += v.x + v.y; }
        }
    }
    if (acc == 12345.678) out[0] = acc;
}

int main()
{
    const long nslots = 1 << 20;
    const int rows = 64;              // 64 rows x 8 doubles = 4 KB per slot -> 4 GB total
    size_t bytes = (size_t)nslots * rows * 8 * 8;
    double *p, *out; hipMalloc(&p, bytes); hipMalloc(&out, 64); hipMemset(p, 0, bytes);
    std::vector<int> sp; srand(1);
    for (long s = 0; s < nslots; ++s) if (rand() % 8 == 0) sp.push_back((int)s);
    int* dsp; hipMalloc(&dsp, sp.size() * 4); hipMemcpy(dsp, sp.data(), sp.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name, const int* list, long nact) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3((nact + 255) / 256), dim3(256), 0, 0, p, nslots, rows, list, nact, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double useful = (double)nact * rows * 64;
        printf("%-28s active %8ld  %8.3f ms  useful %8.1f GB/s\n", name, nact, ms, useful / ms / 1e6);
    };
    run(rd<0>, "SoA 8B dense", nullptr, nslots);
    run(rd<2>, "granule 16B dense", nullptr, nslots);
    run(rd<4>, "granule 32B dense", nullptr, nslots);
    run(rd<8>, "granule 64B dense", nullptr, nslots);
    run(rd<16>, "granule 128B dense", nullptr, nslots);
    run(rd<0>, "SoA 8B sparse 1/8", dsp, (long)sp.size());
    run(rd<2>, "granule 16B sparse 1/8", dsp, (long)sp.size());
    run(rd<4>, "granule 32B sparse 1/8", dsp, (long)sp.size());
    run(rd<8>, "granule 64B sparse 1/8", dsp, (long)sp.size());
    run(rd<16>, "granule 128B sparse 1/8", dsp, (long)sp.size());
    return 0;
}
