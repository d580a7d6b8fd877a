#!/bin/bash
# Cross-compiles a variant of the library here (no GPU needed) into rodygs_amd/csrc/variants/<name>.so: the named object
# is compiled with the given macros INTO variants/ (the regular objects of the tree are never replaced, so a failed or
# interrupted variant build cannot leak an ablation object into the regular library) and linked with the other regular
# objects.  The variant travels to the GPU box with the snapshot; select it with RDG_LIB_PATH (scripts/ab.sh).
#   usage: scripts/build_variant.sh <name> <object, e.g. rdg_deform> "<-D...>"
set -e
cd "$(dirname "$0")/../rodygs_amd/csrc"
mkdir -p variants
make -j8 > /dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
COMMON="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-pass-failed -I../../include"
case "$2" in
  rdg_preprocess_fwd|rdg_binning) MODE="-ffp-contract=off" ;;
  rdg_render|rdg_loss) MODE="-ffp-contract=fast -fno-slp-vectorize" ;;
  *) MODE="-ffp-contract=fast" ;;
esac
$HIPCC $COMMON $MODE $3 -c $2.hip -o variants/$1.$2.o
OTHERS=$(ls *.o | grep -v "^$2.o$")
$HIPCC -shared -fPIC --offload-arch=gfx950 $OTHERS variants/$1.$2.o -ldl -o variants/$1.so
rm -f variants/$1.$2.o
echo "built variants/$1.so ($2 with $3)"
