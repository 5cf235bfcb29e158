// read_sweep.hip -- what read bandwidth can a plain streaming kernel reach on this box, by loads in flight per lane
// (U x 16 B), register sets (1 or 2), workgroups per CU and cache policy?  The ceiling the ExSUM kernel is held against.
// build: hipcc --offload-arch=gfx950 -O3 -o read_sweep read_sweep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT, bool TWOSETS>
__global__ void __launch_bounds__(256) k_read(const double *a, long long n, double *sink)
{
    const d2 *v = (const d2 *)a;
    const long long nv = n >> 1, tile = 256ll * U, ntiles = nv / tile;
    double s = 0;
    auto ld = [&](const d2 *p) { return NT ? __builtin_nontemporal_load(p) : *p; };
    if (!TWOSETS) {
        for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
            d2 r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = ld(v + t * tile + threadIdx.x + u * 256);
#pragma unroll
            for (int u = 0; u < U; ++u) s += r[u].x + r[u].y;
        }
    } else {
        long long t = blockIdx.x;
        d2 r0[U], r1[U];
        auto fill = [&](long long tt, d2 (&r)[U]) {
            const long long q = tt < ntiles ? tt : ntiles - 1;
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = ld(v + q * tile + threadIdx.x + u * 256);
        };
        auto use = [&](d2 (&r)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) s += r[u].x + r[u].y;
        };
        if (t < ntiles) {
            fill(t, r0);
            for (;;) {
                fill(t + gridDim.x, r1); use(r0); t += gridDim.x; if (t >= ntiles) break;
                fill(t + gridDim.x, r0); use(r1); t += gridDim.x; if (t >= ntiles) break;
            }
        }
    }
    if (s == 1.2345e-300) *sink = s;
}
int main()
{
    const long long n = 1ll << 28;
    const int NB = 4;
    double *buf[NB], *sink;
    for (int i = 0; i < NB; ++i) { hipMalloc(&buf[i], n * 8); hipMemset(buf[i], 0x11 * (i + 1), n * 8); }
    hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int grid, const char *name) {
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf[i % NB], n, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int reps = 60;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf[i % NB], n, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s grid %5d: %.1f GB/s\n", name, grid, reps * n * 8.0 / ms / 1e6);
        fflush(stdout);
    };
    for (int bpc : {1, 2, 3, 4, 8, 16}) {
        const int g = 256 * bpc;
        run(k_read<4, true, false>, g, "U4 nt 1set");
        run(k_read<4, true, true>, g, "U4 nt 2sets");
        run(k_read<8, true, false>, g, "U8 nt 1set");
        run(k_read<8, true, true>, g, "U8 nt 2sets");
        run(k_read<4, false, true>, g, "U4 plain 2sets");
        run(k_read<2, true, true>, g, "U2 nt 2sets");
    }
    return 0;
}
