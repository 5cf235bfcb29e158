// blas3.hip -- ExGEMM (placeholder until the kernels land)
#include "exblas_internal.h"
namespace exb {
hipError_t exgemm_dispatch(Ctx &, char, char, int, int, int, double, const double *, int, const double *, int,
                           double, double *, int, int, int, int, hipStream_t)
{
    return hipErrorNotSupported;
}
}  // namespace exb
