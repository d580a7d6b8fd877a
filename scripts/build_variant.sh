#!/bin/bash
# Cross-compiles a variant of the library here (no GPU needed) into rodygs_amd/csrc/variants/<name>.so: the named object
# is rebuilt with the given macros, linked with the regular objects, and the regular build is restored.  The variant
# travels to the GPU box with the snapshot; select it with RDG_LIB_PATH (scripts/ab.sh).
#   usage: scripts/build_variant.sh <name> <object, e.g. rdg_deform> "<-D...>"
set -e
cd "$(dirname "$0")/../rodygs_amd/csrc"
mkdir -p variants
make -j8 > /dev/null
cp $2.o /tmp/$2.keep.o
rm -f $2.o
make EXTRA="$3" $2.o > /dev/null
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 *.o -ldl -o variants/$1.so
cp /tmp/$2.keep.o $2.o
touch $2.o
echo "built variants/$1.so ($2 with $3)"
