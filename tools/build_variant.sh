#!/bin/bash
# Build a tuning variant of libgunrock.so with extra -D defines: bash tools/build_variant.sh <name> "<defines>"
# -> tools/variants/<name>.so ; run with GUNROCK_LIB_PATH=tools/variants/<name>.so
set -e
name=$1; defs=$2
root=$(cd $(dirname $0)/.. && pwd)
src=$root/gunrockinst_amd/csrc
mkdir -p $root/tools/variants $src/build_$name
pids=""
for f in $src/lib/*.hip; do
  o=$src/build_$name/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $defs -I$src -I$root/include -c $f -o $o 2> $o.log &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $src/build_$name/*.o -o $root/tools/variants/$name.so
echo built $root/tools/variants/$name.so
