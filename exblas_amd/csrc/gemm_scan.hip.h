// gemm_scan.hip.h -- operand scan shared by the two error-free-slicing ExGEMM paths (blas3_mfma.hip: 21-bit fp64
// slices on v_mfma_f64_16x16x4_f64; blas3_i8.hip: 8-bit slices on v_mfma_i32_32x32x32_i8).
// Per vector (row of A' = fl(alpha*A), column of B): the scale exponent (|x| < 2^ea) and the lowest set bit; globally:
// how many bits a vector spans at most (-> slice count), the exponent range, and whether anything is non-finite or
// subnormal (-> scalar kernel).
#pragma once
#include <hip/hip_runtime.h>

namespace exb {

// info words written by the scan kernels
enum { INFO_NEED_A = 0, INFO_NEED_B = 1, INFO_FLAGS = 2, INFO_EMIN = 3, INFO_EMAX = 4, INFO_WORDS = 16 };

// ---------------------------------------------------------------------------------------------
// scan: per vector (row of A' / column of B) the scale, and globally the number of bits to cover
// ---------------------------------------------------------------------------------------------
struct ScanAcc {
    int emax, lsbmin;
    unsigned bad;
    __device__ __forceinline__ void init() { emax = -100000; lsbmin = 100000; bad = 0; }
    __device__ __forceinline__ void add(double x)
    {
        const unsigned long long u = (unsigned long long)__double_as_longlong(x);
        const unsigned be = (unsigned)(u >> 52) & 0x7ffu;
        const unsigned long long frac = u & 0x000fffffffffffffull;
        if (be == 0) {
            if (frac) bad = 1;  // subnormal input: scalar path
            return;             // zero
        }
        if (be == 0x7ffu) { bad = 1; return; }
        const int e = (int)be - 1023;
        const unsigned long long mant = frac | 0x0010000000000000ull;
        const int lsb = e - 52 + __builtin_ctzll(mant);
        emax = max(emax, e);
        lsbmin = min(lsbmin, lsb);
    }
    __device__ __forceinline__ void merge(const ScanAcc &o)
    {
        emax = max(emax, o.emax);
        lsbmin = min(lsbmin, o.lsbmin);
        bad |= o.bad;
    }
};

// Both scan kernels only fold their part of a vector into vmax[v] / vlsb[v] (atomicMax / atomicMin), so the
// reduction dimension can be split over workgroups; k_scan_finish then derives the scale and the global needs.
__device__ __forceinline__ void scan_publish(const ScanAcc &s, int *vmax, int *vlsb, int *info)
{
    if (s.emax > -50000) {
        atomicMax(vmax, s.emax);
        atomicMin(vlsb, s.lsbmin);
    }
    if (s.bad) atomicOr((unsigned *)&info[INFO_FLAGS], 1u);
}

// vectors whose elements are contiguous (stride 1 along the reduction): one workgroup per vector
static __global__ void __launch_bounds__(256) k_scan_contig(const double *__restrict__ p, long long ldv, int nvec, int len,
                                                     double scale, int *vmax, int *vlsb, int *info)
{
    __shared__ ScanAcc red[256];
    const int v = blockIdx.x;
    if (v >= nvec) return;
    ScanAcc s;
    s.init();
    const double *q = p + (long long)v * ldv;
    for (int i = threadIdx.x; i < len; i += 1024) {  // four loads in flight per thread
        double w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = i + 256 * u < len ? q[i + 256 * u] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) s.add(scale * w[u]);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x].merge(red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) scan_publish(red[0], &vmax[v], &vlsb[v], info);
}

// vectors whose elements are strided by ldv (adjacent vectors are contiguous): one thread per vector and
// per slice of the reduction dimension (blockIdx.y)
static __global__ void __launch_bounds__(256) k_scan_strided(const double *__restrict__ p, long long ldv, int nvec, int len,
                                                      double scale, int *vmax, int *vlsb, int *info)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nvec) return;
    const int per = (len + gridDim.y - 1) / gridDim.y;
    const int i0 = blockIdx.y * per, i1 = min(len, i0 + per);
    ScanAcc s;
    s.init();
    for (int i = i0; i < i1; i += 4) {  // four loads in flight per thread
        double w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = i + u < i1 ? p[(long long)(i + u) * ldv + v] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) s.add(scale * w[u]);
    }
    scan_publish(s, &vmax[v], &vlsb[v], info);
}

// also resets the info block (first thread), so the whole scan is kernels only: capturable, no host memory involved
static __global__ void __launch_bounds__(256) k_scan_init(int nvec, int *vmax, int *vlsb, int *info)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < nvec) {
        vmax[v] = -100000;
        vlsb[v] = 100000;
    }
    if (v < INFO_WORDS) info[v] = v == INFO_EMIN ? 100000 : (v == INFO_EMAX ? -100000 : 0);
}

// vmax -> scale ea = emax + 1 (in place), and the global slice need / exponent range
static __global__ void __launch_bounds__(256) k_scan_finish(int nvec, int *vmax, const int *vlsb, int *info, int need_slot)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nvec) return;
    const int e = vmax[v];
    if (e < -50000) {
        vmax[v] = 0;
        return;
    }
    vmax[v] = e + 1;
    atomicMax(&info[need_slot], e + 1 - vlsb[v]);
    atomicMin(&info[INFO_EMIN], e);
    atomicMax(&info[INFO_EMAX], e);
}

}  // namespace exb
