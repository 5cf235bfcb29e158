// config.h -- static stand-in for the file the reference generates from cmake/config.h.in.
// The HIP backend needs no source/binary directory (there is no runtime kernel JIT).
#ifndef EXBLAS_CONFIG_H_
#define EXBLAS_CONFIG_H_
#define EXBLAS_VERSION_MAJOR 1
#define EXBLAS_VERSION_MINOR 0
#define EXBLAS_SOURCE_DIR ""
#define EXBLAS_BINARY_DIR ""
#define USE_EXBLAS
#define EXBLAS_GPU_HIP 1
#endif
