#!/bin/bash
# A/B builds of the library with extra macros for ONE source file:
#   tools/ab_build.sh <tag> <source.hip> [-DNAME=VALUE ...]   ->  exblas_amd/lib/ab/libexblas_<tag>.so
# The other objects are compiled once into /tmp/exblas_ab_objs and reused.  Load with EXBLAS_AMD_LIB=<path>.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; SRC=$2; shift 2
OBJ=/tmp/exblas_ab_objs; mkdir -p $OBJ $ROOT/exblas_amd/lib/ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I$ROOT/include"
for s in blas1.hip blas2.hip trsv.hip blas3.hip blas3_mfma.hip blas3_i8.hip blas3_crt.hip capi.hip comm.hip generators.cpp; do
  o=$OBJ/${s//./_}.o
  if [ "$s" != "$SRC" ] && { [ ! -f $o ] || [ $ROOT/exblas_amd/csrc/$s -nt $o ]; }; then
    hipcc $FLAGS -x hip -c $ROOT/exblas_amd/csrc/$s -o $o &
  fi
done
hipcc $FLAGS "$@" -x hip -c $ROOT/exblas_amd/csrc/$SRC -o $OBJ/${SRC//./_}.$TAG.o
wait
objs=""
for s in blas1.hip blas2.hip trsv.hip blas3.hip blas3_mfma.hip blas3_i8.hip blas3_crt.hip capi.hip comm.hip generators.cpp; do
  if [ "$s" = "$SRC" ]; then objs="$objs $OBJ/${s//./_}.$TAG.o"; else objs="$objs $OBJ/${s//./_}.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/exblas_amd/lib/ab/libexblas_$TAG.so $objs
echo $ROOT/exblas_amd/lib/ab/libexblas_$TAG.so
