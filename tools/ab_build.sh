#!/bin/bash
# Variant library for a same-box A/B: tools/ab_build.sh <name> "<extra flags>" file1.hip file2.hip ...
# compiles only the named translation units with the extra flags (objects under /tmp) and links them with the in-tree objects
# of the others into tools/ab/lib_<name>.so (git-ignored; travels to the GPU box; see tools/ab_bench.sh).
set -e
name=$1; extra=$2; shift 2
cd "$(dirname "$0")/../effocr_amd/csrc"
mkdir -p ../../tools/ab /tmp/ab_$name; rm -f /tmp/ab_$name/*.o
objs=""
for o in *.o; do
  src=${o%.o}.hip
  case $o in panel.o|mlp_bf16.o|mlp_f16.o) continue;; esac   # A/B-only objects (make AB=1): the variant is a PRODUCT library (ab_stubs.o stands in)
  if [[ " $* " == *" $src "* ]]; then
    flags=""; [ "$src" = qkvattn.hip ] && flags="-mllvm -amdgpu-mfma-vgpr-form"
    [[ "$src" == mlp*.hip ]] && flags="-fno-slp-vectorize"
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags $extra -c $src -o /tmp/ab_$name/$o 2>/tmp/ab_$name/${o%.o}.log &
    objs="$objs /tmp/ab_$name/$o"
  else
    objs="$objs $o"
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/ab/lib_$name.so $objs
ls -la ../../tools/ab/lib_$name.so
