#!/bin/bash
# Build an experiment variant of the library WITHOUT touching the shipping sources: copy aeonflux_amd/csrc + include to a
# scratch tree, apply a patch from tools/experiments/ (switches that must not live in the product's hot loops), compile with
# the given -D flags, and drop the result in variants/<name>.so (git-ignored; travels to the GPU box) for tools/ab_bench.sh
# and tools/traffic_experiments.sh.
#   tools/build_variant.sh <name> <patch|-> [-DAFX_EXPERIMENT_...]...
# The patches are taken against a named commit of kernels.hip and may need refreshing when the kernels move on.
set -eu
NAME=$1; PATCH=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/afx_variant.XXXXXX)
mkdir -p $T/aeonflux_amd $T/include $R/variants
cp -r $R/aeonflux_amd/csrc $T/aeonflux_amd/csrc
cp $R/include/*.h $T/include/
rm -rf $T/aeonflux_amd/csrc/build
if [ "$PATCH" != "-" ]; then (cd $T && patch -p1 < $R/$PATCH); fi
EXTRA="$*"
(cd $T/aeonflux_amd/csrc && sed -i "s|-O3 --offload-arch|-O3 $EXTRA --offload-arch|; s|^HOSTFLAGS = |HOSTFLAGS = $EXTRA |" Makefile && make -s ARCH=gfx950)
cp $T/aeonflux_amd/lib/libaeonflux_gpu.so $R/variants/$NAME.so
rm -rf $T
echo "variants/$NAME.so"
