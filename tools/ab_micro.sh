#!/bin/bash
# tools/ab_micro.sh LIB_A LIB_B "shape shape ..." [mode] [env...]: tools/conv_micro.py on each shape with two builds of the library, same box,
# alternated twice (A B A B); LIB = "-" for the default library.
a=$1; b=$2; shapes=$3; mode=${4:-fwd}
for sh in $shapes; do
  for rep in 1 2; do
    for lib in "$a" "$b"; do
      if [ "$lib" = "-" ]; then unset MRFP_HIP_LIB; else export MRFP_HIP_LIB=$lib; fi
      echo -n "[$lib] "; python tools/conv_micro.py $sh 30 $mode 2>&1 | tail -1
    done
  done
done
