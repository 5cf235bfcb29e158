// lone_wave.hip -- how fast does ONE wavefront retire dependent / independent fp64 instructions on this GPU, and at
// what shader clock?  hipcc --offload-arch=gfx950 -O3 -o lone_wave lone_wave.hip && ./lone_wave
// s_memtime (clock64) counts shader clocks, wall_clock64 a constant 100 MHz reference.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_chain(double *out, long long *t, int iters, int mode)
{
    double a = out[0], b = out[1], c = out[2], d = out[3];
    const long long w0 = wall_clock64(), c0 = clock64();
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) {  // one dependent chain: 8 adds per iteration
            a = a + b; a = a + b; a = a + b; a = a + b;
            a = a + b; a = a + b; a = a + b; a = a + b;
        }
    } else {
        for (int i = 0; i < iters; ++i) {  // four independent chains: 8 adds per iteration
            a = a + 1.0; b = b + 1.0; c = c + 1.0; d = d + 1.0;
            a = a + 1.0; b = b + 1.0; c = c + 1.0; d = d + 1.0;
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[4 + threadIdx.x % 4] = a + b + c + d;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}

int main()
{
    double *out; long long *t;
    hipMalloc(&out, 64 * sizeof(double)); hipMalloc(&t, 2 * sizeof(long long));
    double h[8] = {1.0, 1e-30, 1.0, 1.0, 0, 0, 0, 0};
    hipMemcpy(out, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 200000;
    for (int blocks : {1, 256, 2048}) {
        for (int mode = 0; mode < 2; ++mode) {
            long long ht[2];
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64), 0, 0, out, t, iters, mode);
                hipDeviceSynchronize();
            }
            hipMemcpy(ht, t, sizeof(ht), hipMemcpyDeviceToHost);
            const double ns = ht[1] * 10.0, instr = 8.0 * iters;
            printf("blocks=%4d (1 wave each) %s: %.2f shader clocks/instr, %.2f ns/instr, shader clock %.0f MHz\n", blocks,
                   mode ? "4 independent chains" : "1 dependent chain    ", ht[0] / instr, ns / instr, ht[0] / ns * 1e3);
        }
    }
    return 0;
}
