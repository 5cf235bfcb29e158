// test_comm.cpp -- stand-alone C++ caller of the library's multi-GPU entry points (include/exblas_hip.h, section 2b):
// what a C++ / MPI user of the reference's distributed ExSUM (src/cpu/blas/blas1/ExSUM.cpp:33-63,142-152) links against.
// No Python, no torch, no MPI here: one rank, once over the RCCL transport (ncclGetUniqueId / ncclCommInitRank /
// ncclAllReduce / ncclBroadcast issued by libexblas.so; EXBLAS_COMM_FORCE=1 makes the sharded calls issue their
// broadcasts and all-gathers although there is only one rank) and once over the host-callback transport -- both must
// reproduce the plain single-GPU calls bit for bit.
#include "exblas_hip.h"
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { int rc_ = (int)(x); if (rc_ != 0) { printf("%s -> %d\n", #x, rc_); return 1; } } while (0)

static int cb_allreduce(void *, int64_t *, int64_t) { return 0; }            // one rank: the sum is the value itself
static int cb_bcast(void *, void *, int64_t, int) { return 0; }
static int cb_allgatherv(void *, void *, const int64_t *) { return 0; }

int main()
{
    setenv("EXBLAS_COMM_FORCE", "1", 1);
    CK(exblas_hip_init(-1));
    const int64_t n = (1 << 22) + 3;
    const int m = 2300, nn = 70, k = 150;
    double *x, *y, *A, *B, *C, *C2, *yv, *yv2;
    int64_t *rec, *rec2;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8));
    CK(hipMalloc(&A, (size_t)m * k * 8)); CK(hipMalloc(&B, (size_t)k * nn * 8));
    CK(hipMalloc(&C, (size_t)m * nn * 8)); CK(hipMalloc(&C2, (size_t)m * nn * 8));
    CK(hipMalloc(&yv, (size_t)m * 8)); CK(hipMalloc(&yv2, (size_t)m * 8));
    CK(hipMalloc(&rec, EXBLAS_OUT_WORDS * 8)); CK(hipMalloc(&rec2, EXBLAS_OUT_WORDS * 8));
    CK(exblas_gen_dev(EXBLAS_GEN_ILLCOND, 1, 0, n, n, 1e32, 0, x, nullptr));
    CK(exblas_gen_dev(EXBLAS_GEN_LOGNORMAL, 2, 0, n, n, 0.0, 2.0, y, nullptr));
    CK(exblas_gen_dev(EXBLAS_GEN_FPUNIFORM_SIGNED, 3, 0, (int64_t)m * k, (int64_t)m * k, 20, 10, A, nullptr));
    CK(exblas_gen_dev(EXBLAS_GEN_FPUNIFORM_SIGNED, 4, 0, (int64_t)k * nn, (int64_t)k * nn, 20, 10, B, nullptr));
    bool pass = true;
    for (int transport = 0; transport < 2; ++transport) {
        exblas_comm_t *comm = nullptr;
        if (transport == 0) {
            unsigned char id[EXBLAS_UNIQUE_ID_BYTES];
            CK(exblas_comm_unique_id(id));
            CK(exblas_comm_init_rccl(&comm, 1, 0, id));
        } else {
            CK(exblas_comm_init_host(&comm, 1, 0, cb_allreduce, cb_bcast, cb_allgatherv, nullptr));
        }
        if (exblas_comm_size(comm) != 1 || exblas_comm_rank(comm) != 0) pass = false;
        int64_t f, l;
        exblas_shard_range(n, 0, 1, &f, &l);
        if (f != 0 || l != n) pass = false;
        // ExSUM / ExDOT
        int64_t h1[EXBLAS_OUT_WORDS], h2[EXBLAS_OUT_WORDS];
        CK(exblas_exsum_dev(x, n, 1, 8, 1, nullptr, rec));
        CK(exblas_exsum_allreduce_dev(comm, x, n, 1, 8, 1, nullptr, rec2));
        CK(hipMemcpy(h1, rec, sizeof h1, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2, rec2, sizeof h2, hipMemcpyDeviceToHost));
        if (memcmp(h1, h2, sizeof(int64_t) * (EXBLAS_OUT_CANON + EXBLAS_NCANON))) { pass = false; printf("exsum differs (%d)\n", transport); }
        CK(exblas_exdot_dev(x, 1, y, 1, n, 8, 1, nullptr, rec));
        CK(exblas_exdot_allreduce_dev(comm, x, 1, y, 1, n, 8, 1, nullptr, rec2));
        CK(hipMemcpy(h1, rec, sizeof h1, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2, rec2, sizeof h2, hipMemcpyDeviceToHost));
        if (memcmp(h1, h2, sizeof(int64_t) * (EXBLAS_OUT_CANON + EXBLAS_NCANON))) { pass = false; printf("exdot differs (%d)\n", transport); }
        double d; memcpy(&d, &h2[EXBLAS_OUT_EXACT], 8);
        printf("  transport %d: exdot = %.17g\n", transport, d);
        // ExGEMM, rows in 4 chunks with the RCCL transport
        CK(hipMemset(C, 0, (size_t)m * nn * 8)); CK(hipMemset(C2, 0, (size_t)m * nn * 8));
        CK(exblas_exgemm_dev('N', 'N', m, nn, k, 1.0, A, k, B, nn, 0.0, C, nn, 8, 1, nullptr));
        CK(exblas_exgemm_sharded_dev(comm, 'N', 'N', m, nn, k, 1.0, A, k, B, nn, 0, 0.0, C2, nn, /*gather*/ 1, 8, 1, nullptr));
        std::vector<double> c1((size_t)m * nn), c2((size_t)m * nn);
        CK(hipMemcpy(c1.data(), C, c1.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), C2, c2.size() * 8, hipMemcpyDeviceToHost));
        if (memcmp(c1.data(), c2.data(), c1.size() * 8)) { pass = false; printf("exgemm differs (%d)\n", transport); }
        // ExGEMV 'T' on the same storage read column-major (k x m matrix with lda = k: outputs = its m columns)
        CK(hipMemset(yv, 0, (size_t)m * 8)); CK(hipMemset(yv2, 0, (size_t)m * 8));
        CK(exblas_exgemv_dev('T', k, m, 1.0, A, k, B, 1, 0.0, yv, 1, 8, 1, nullptr));
        CK(exblas_exgemv_sharded_dev(comm, 'T', k, m, 1.0, A, k, B, 1, 0, 0.0, yv2, 1, /*gather*/ 1, 8, 1, nullptr));
        std::vector<double> y1(m), y2(m);
        CK(hipMemcpy(y1.data(), yv, m * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(y2.data(), yv2, m * 8, hipMemcpyDeviceToHost));
        if (memcmp(y1.data(), y2.data(), m * 8)) { pass = false; printf("exgemv differs (%d)\n", transport); }
        CK(exblas_comm_destroy(comm));
    }
    printf(pass ? "TestPassed; ALL OK!\n" : "TestFailed!\n");
    return pass ? 0 : 1;
}
